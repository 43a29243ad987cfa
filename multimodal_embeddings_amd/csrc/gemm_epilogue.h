// Fused GEMM epilogues shared by the 128x128 and 256x256 MFMA kernels.
//
// A lane always owns 4 consecutive output columns n..n+3 of one row m (operands are fed to
// the MFMA swapped so that the accumulator's register index runs along n): bias / residual /
// position vectors load as one 16- or 8-byte vector and the result stores as bf16x4 (8 B) or
// f32x4 (16 B).
#pragma once
#include <type_traits>

#include "common.h"
#include "kernels.h"

// erf-GELU (transformers ACT2FN["gelu"], modeling_vit.py:241-255): gelu(x) = x * Phi(x).
// The fc1 epilogue evaluates 32768 of these per wave tile and is VALU-issue bound, so the form
// with the fewest issues is used: Phi(x) = sigmoid(x * (c0 + c1 x^2 + c2 x^4)), an odd-polynomial
// fit of logit(Phi) (minimax over |x| <= 9, coefficients fitted against 0.5*(1+erf(x/sqrt2)) in
// f64; x^2 is clamped at 81 where the quartic term would turn over).  max |gelu - erf-gelu| =
// 2.6e-5 absolute over the whole real line -- two orders below the bf16 rounding (2^-9 relative)
// applied to the result right after.  7 plain VALU ops + exp2 + rcp, branch-free.
__device__ __forceinline__ float gelu_erf(float x) {
    const float x2 = fminf(x * x, 81.0f);
    float q = fmaf(x2, -0.0007030335771975101f * -1.4426950408889634f, 0.07401129204501875f * -1.4426950408889634f);
    q = fmaf(q, x2, 1.595015768572222f * -1.4426950408889634f);
    const float e = __builtin_amdgcn_exp2f(x * q);  // exp(-p(x))
    return x * __builtin_amdgcn_rcpf(1.0f + e);  // v_rcp_f32 (1 ulp); __frcp_rn would expand to a full IEEE division
}

// ---- LayerNorm statistics in ONE canonical summation order ---------------------------------------------
// Shared by the EPI_BIAS_RES_STATS epilogue (partial sums while the rounded outputs are still in registers) and
// by the stand-alone kernels of rowops.hip, so that an embedding does not depend on which of them ran:
//   per 64-column slice, four column groups g = 0..3 (columns 16 j + 4 g + r, j and r ascending), each summed by
//   v_dot2c_f32_bf16 over its packed pairs in that order; slice total ((g0 + g1) + (g2 + g3)); row totals over
//   the slices sequentially in f64; mean = S / d, var = Q / d - mean^2 (>= 0), rstd = 1 / sqrt(var + eps).
typedef __attribute__((ext_vector_type(2))) __bf16 ln_bf16x2;
__device__ __forceinline__ void ln_accumulate(uint2 pk, float& s, float& q) {
    const ln_bf16x2 one = {(__bf16)1.0f, (__bf16)1.0f};
    const ln_bf16x2 a = __builtin_bit_cast(ln_bf16x2, pk.x), b = __builtin_bit_cast(ln_bf16x2, pk.y);
    s = __builtin_amdgcn_fdot2_f32_bf16(a, one, s, false);
    q = __builtin_amdgcn_fdot2_f32_bf16(a, a, q, false);
    s = __builtin_amdgcn_fdot2_f32_bf16(b, one, s, false);
    q = __builtin_amdgcn_fdot2_f32_bf16(b, b, q, false);
}
__device__ __forceinline__ float2 ln_finish_row(double S, double Q, int d, float eps) {
    const double mean = S / d;
    double var = Q / d - mean * mean;
    var = var > 0.0 ? var : 0.0;
    return make_float2((float)mean, (float)(1.0 / sqrt(var + (double)eps)));
}

struct EpiRow {
    int64_t orow;  // output row (EPI_PATCH remaps patch rows past the [CLS] rows)
    int prow;      // position row for EPI_PATCH
};

template <int EPI>
__device__ __forceinline__ EpiRow epi_row(int m) {
    EpiRow r{m, 0};
    if (EPI == EPI_PATCH) {
        const int b = m / VIT_NP;
        r.prow = m - b * VIT_NP + 1;
        r.orow = (int64_t)b * VIT_T + r.prow;
    }
    return r;
}

// v = accumulator values for (m, n..n+3); n < N and n % 4 == 0 guaranteed by the caller.
template <int EPI>
__device__ __forceinline__ void epi_store(const GemmArgs& g, int m, const EpiRow& er, int n, f32x4 v) {
    if (EPI == EPI_TOPK) {
        const float tau = g.thr[(int64_t)m * g.thr_stride];
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (n + r < g.N && v[r] >= tau) {
                const int k = atomicAdd(g.cand_count + m, 1);
                if (k < g.cand_cap) {
                    g.cand_val[(int64_t)m * g.cand_cap + k] = v[r];
                    g.cand_idx[(int64_t)m * g.cand_cap + k] = n + r;
                } else {
                    atomicOr(g.overflow, 1);
                }
            }
        return;
    }
    if (EPI == EPI_F32) {
        float* o = g.outf + (int64_t)m * g.ldf + n;
        if (n + 3 < g.N && ((g.ldf & 3) == 0)) {
            *(f32x4*)o = v;
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (n + r < g.N) o[r] = v[r];
        }
        return;
    }
    if (EPI == EPI_LN_BIAS || EPI == EPI_LN_BIAS_GELU) {
        const float mu = g.ln_stats[2 * (int64_t)m], rs = g.ln_stats[2 * (int64_t)m + 1];
        const f32x4 sv = *(const f32x4*)(g.colsum + n), bv = *(const f32x4*)(g.bias + n);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = fmaf(rs, fmaf(-mu, sv[r], v[r]), bv[r]);
    } else {
        v += *(const f32x4*)(g.bias + n);
    }
    if (EPI == EPI_BIAS_GELU || EPI == EPI_LN_BIAS_GELU) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = gelu_erf(v[r]);
    }
    if (EPI == EPI_PATCH) v += *(const f32x4*)(g.pos + (int64_t)er.prow * g.N + n);
    bf16_t* o = (bf16_t*)g.out + er.orow * g.ldo + n;
    if (EPI == EPI_BIAS_RES || EPI == EPI_BIAS_RES_STATS) {  // (partial sums of edge tiles: the host re-derives those rows)
        const bf16x4 rv = *(const bf16x4*)((const bf16_t*)g.res + er.orow * g.ldo + n);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] += (float)rv[r];
    }
    bf16x4 ov;
#pragma unroll
    for (int r = 0; r < 4; ++r) ov[r] = (bf16_t)v[r];
    *(bf16x4*)o = ov;
}

// ---- 16-byte epilogue accesses (halves the number of store / load instructions) -------------
// The epilogue of a 256x256 tile is store-ISSUE bound (one CU retires a vector store
// instruction every few tens of cycles whatever its width), so width is what counts.  A lane
// owns 4 consecutive columns (8 bytes of bf16) of a 16x16 accumulator tile; lanes fq and fq^1
// (16 lanes apart) own the neighbouring 4.  v_permlane16_swap exchanges the odd 16-lane rows of
// one register with the even rows of another: applied to the packed data of two adjacent
// column tiles (j, j+1) it leaves every lane with 8 consecutive columns of ONE tile:
//   fq even -> tile j,   columns 8*(fq>>1) .. +7      fq odd -> tile j+1, same columns
// The same swap applied to 16 bytes loaded in that layout returns them to accumulator layout.
__device__ __forceinline__ uint4 pair_to_row16(uint2 p, uint2 q) {
    const auto a = __builtin_amdgcn_permlane16_swap(p.x, q.x, false, false);
    const auto b = __builtin_amdgcn_permlane16_swap(p.y, q.y, false, false);
    return make_uint4(a[0], b[0], a[1], b[1]);
}
__device__ __forceinline__ void row16_to_pair(uint4 x, uint2& p, uint2& q) {
    const auto a = __builtin_amdgcn_permlane16_swap(x.x, x.z, false, false);
    const auto b = __builtin_amdgcn_permlane16_swap(x.y, x.w, false, false);
    p = make_uint2(a[0], b[0]);
    q = make_uint2(a[1], b[1]);
}
__device__ __forceinline__ uint2 pack_bf16x4(f32x4 v) {
    bf16x4 o;
#pragma unroll
    for (int r = 0; r < 4; ++r) o[r] = (bf16_t)v[r];
    return __builtin_bit_cast(uint2, o);
}
__device__ __forceinline__ f32x4 unpack_bf16x4(uint2 u) {
    const bf16x4 b = __builtin_bit_cast(bf16x4, u);
    return f32x4{(float)b[0], (float)b[1], (float)b[2], (float)b[3]};
}
// column (within the wave's 64) of this lane's 16-byte piece for the tile pair (jp, jp+1)
__device__ __forceinline__ int row16_col(int jp, int fq) { return ((fq & 1) ? (jp + 1) * 16 : jp * 16) + (fq >> 1) * 8; }

// Interior-tile epilogue of one wave tile (128 x 64 as acc[8][4] of 16x16 MFMA tiles; row
// mw + 16 i + fr, columns nw + 16 j + 4 fq ..) for the bias / GELU / residual / LayerNorm-folded
// epilogues.  All loads are issued first, `between()` runs next (the GEMM kernels request their
// next LDS-DMA there: vmcnt retires in order and counts stores, so a load issued after a store
// would wait for it), then 16 fire-and-forget 16-byte stores, which are afterwards exactly the
// 16 youngest vector-memory operations of the wave.  Straight-line code: a branch would make the
// compiler re-insert vmcnt(0) at every join.
// NDEF > 0: the last NDEF / 2 row blocks of the wave tile (i >= 8 - NDEF / 2) are not stored but returned packed in pend[];
// the caller issues them later (gemm256r.hip, variant 4).
template <int EPI, int NDEF = 0, class Between>
__device__ __forceinline__ void epilogue_wave_128x64(const GemmArgs& g, f32x4 (&acc)[8][4], int mw, int nw, int fr, int fq,
                                                     Between&& between, uint4* pend = nullptr) {
    static_assert(EPI == EPI_BIAS || EPI == EPI_BIAS_GELU || EPI == EPI_BIAS_RES || EPI == EPI_LN_BIAS || EPI == EPI_LN_BIAS_GELU ||
                      EPI == EPI_BIAS_RES_STATS,
                  "epilogue_wave_128x64: unsupported epilogue");
    constexpr bool RES = EPI == EPI_BIAS_RES || EPI == EPI_BIAS_RES_STATS;
    constexpr bool STATS = EPI == EPI_BIAS_RES_STATS;
    constexpr bool LN = EPI == EPI_LN_BIAS || EPI == EPI_LN_BIAS_GELU;
    constexpr bool GELU = EPI == EPI_BIAS_GELU || EPI == EPI_LN_BIAS_GELU;
    // wave-uniform row bases (SGPR pairs) + 32-bit lane offsets: saddr addressing
    const int64_t tile_off = (int64_t)mw * g.ldo + nw;
    const int lo0 = fr * (int)g.ldo + row16_col(0, fq), lo1 = fr * (int)g.ldo + row16_col(2, fq);
    const bf16_t* resb = (const bf16_t*)g.res + tile_off;
    bf16_t* outb = (bf16_t*)g.out + tile_off;
    f32x4 bv[4], sv[LN ? 4 : 1];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        bv[j] = *(const f32x4*)(g.bias + nw + fq * 4 + j * 16);
        if (LN) sv[j] = *(const f32x4*)(g.colsum + nw + fq * 4 + j * 16);
    }
    float2 st[LN ? 8 : 1];
    if (LN) {
#pragma unroll
        for (int i = 0; i < 8; ++i) st[i] = *(const float2*)(g.ln_stats + 2 * (int64_t)(mw + i * 16 + fr));
    }
    uint4 rv[RES ? 8 : 1][2];
    if (RES) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            rv[i][0] = *(const uint4*)(resb + (int64_t)i * 16 * g.ldo + lo0);
            rv[i][1] = *(const uint4*)(resb + (int64_t)i * 16 * g.ldo + lo1);
        }
    }
    asm volatile("" ::: "memory");
    between();
    asm volatile("" ::: "memory");
    float psum[STATS ? 8 : 1], psq[STATS ? 8 : 1];
    if (STATS) {
#pragma unroll
        for (int i = 0; i < 8; ++i) psum[i] = psq[i] = 0.f;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
#pragma unroll
        for (int jp = 0; jp < 4; jp += 2) {
            f32x4 v0, v1;
            if (LN) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    v0[r] = fmaf(st[i].y, fmaf(-st[i].x, sv[jp][r], acc[i][jp][r]), bv[jp][r]);
                    v1[r] = fmaf(st[i].y, fmaf(-st[i].x, sv[jp + 1][r], acc[i][jp + 1][r]), bv[jp + 1][r]);
                }
            } else {
                v0 = acc[i][jp] + bv[jp];
                v1 = acc[i][jp + 1] + bv[jp + 1];
            }
            if (GELU) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    v0[r] = gelu_erf(v0[r]);
                    v1[r] = gelu_erf(v1[r]);
                }
            }
            if (RES) {
                uint2 rp, rq;
                row16_to_pair(rv[i][jp >> 1], rp, rq);
                v0 += unpack_bf16x4(rp);
                v1 += unpack_bf16x4(rq);
            }
            const uint2 pk0 = pack_bf16x4(v0), pk1 = pack_bf16x4(v1);
            if (STATS) {  // canonical order: column tiles j ascending, the two packed pairs of a tile in order
                ln_accumulate(pk0, psum[i], psq[i]);
                ln_accumulate(pk1, psum[i], psq[i]);
            }
            const uint4 packed = pair_to_row16(pk0, pk1);
            if (NDEF > 0 && i >= 8 - NDEF / 2) {
                pend[(i - (8 - NDEF / 2)) * 2 + (jp >> 1)] = packed;
            } else {
                *(uint4*)(outb + (int64_t)i * 16 * g.ldo + (jp ? lo1 : lo0)) = packed;
            }
        }
    }
    if (STATS) {
        // 16 per-lane partials (8 row blocks x {sum, sum of squares}) over this lane's 16 columns of the wave's
        // 64-column slice -> totals over the four lanes (fq = 0..3) that share a row, as ((fq0 + fq1) + (fq2 + fq3)).
        // Two swap levels reduce all 16 at once: v_permlane16_swap leaves the pair sums of quantity a in the even
        // 16-lane rows and those of b in the odd rows, v_permlane32_swap then pairs the lane halves.
        float t[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {  // row block k: rows 0 / 2 of the result pair psum over fq, rows 1 / 3 psq
            const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(psum[k]), __float_as_uint(psq[k]), false, false);
            t[k] = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);  // even rows: psum[k] over (fq, fq + 1); odd rows: psq[k]
        }
        float u[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(t[2 * k]), __float_as_uint(t[2 * k + 1]), false, false);
            u[k] = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);  // lanes 0..31: row block 2k; lanes 32..63: row block 2k + 1
        }
        // lane (fq, fr): fq & 1 selects the plane (0 sum, 1 sum of squares), fq >> 1 the row block 2k + (fq >> 1)
        float* plane = g.ln_part + ((int64_t)(fq & 1) * (g.N >> 6) + (nw >> 6)) * g.ln_part_rows + mw + (fq >> 1) * 16 + fr;
#pragma unroll
        for (int k = 0; k < 4; ++k) plane[k * 32] = u[k];
    }
}

// Interior-tile epilogue of one wave tile of the PATCH-EMBED GEMM (EPI_PATCH): row m = (crop b, patch p) of the im2col
// matrix goes to token row b * 197 + 1 + p of the residual stream, + bias[n] + pos[1 + p, n] (f32), rounded to bf16.
// Same rules as epilogue_wave_128x64 -- 16-byte stores built with v_permlane16_swap, loads never behind stores -- with
// two differences: (1) the position rows are f32 and per lane (a 128-row wave tile crosses at most one crop boundary:
// 196 > 128), 32 x 16 bytes per lane, more registers than the accumulators leave free, so they are fetched in two
// halves: the loads of row blocks 4..7 are ISSUED before the stores of row blocks 0..3 (vmcnt retires in order and
// counts stores: waiting for those loads then leaves exactly the eight stores in flight); (2) when g.ln_part is set
// the LayerNorm partial sums of the rounded outputs are left per OUTPUT row in the canonical order (ln_accumulate),
// so that the first LayerNorm of the pass needs no pass over x (the [CLS] rows, written by cls_rows, and the rows of
// a ragged last tile get theirs from ln_stats_canonical_rows).
template <class Between>
__device__ __forceinline__ void epilogue_wave_patch_128x64(const GemmArgs& g, f32x4 (&acc)[8][4], int mw, int nw, int fr, int fq, Between&& between) {
    const int b0 = mw / VIT_NP;                                  // wave-uniform: crop of the tile's first row
    const int64_t orow0 = (int64_t)mw + b0 + 1;                  // its token row
    bf16_t* outb = (bf16_t*)g.out + orow0 * g.ldo + nw;          // wave-uniform base; lane offsets fit 32 bits
    const int c0 = row16_col(0, fq), c1 = row16_col(2, fq);
    f32x4 bv[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) bv[j] = *(const f32x4*)(g.bias + nw + fq * 4 + j * 16);
    // per row block i: does this lane's row lie in the NEXT crop (bit i)?  Output row and position row follow from it.
    const int first_next = (b0 + 1) * VIT_NP - mw - fr;  // rows with 16 i >= first_next belong to crop b0 + 1
    const int lo_base = fr * (int)g.ldo, po_base = (mw - b0 * VIT_NP + 1 + fr) * g.N + nw + fq * 4;
    auto lo_of = [&](int i) { return lo_base + (i * 16 + (i * 16 >= first_next ? 1 : 0)) * (int)g.ldo; };
    auto po_of = [&](int i) { return po_base + (i * 16 - (i * 16 >= first_next ? VIT_NP : 0)) * g.N; };
    const bool stats = g.ln_part != nullptr;  // wave-uniform
    float psum[8], psq[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) psum[i] = psq[i] = 0.f;
    // quarters of two row blocks: the position rows of quarter k + 1 are requested before the stores of quarter k
    f32x4 pv[2][2][4];
    uint4 packed[2][2];
    auto load_pos = [&](auto k_tag) {
        constexpr int K = decltype(k_tag)::value;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) pv[K & 1][i][j] = *(const f32x4*)(g.pos + po_of(2 * K + i) + j * 16);
    };
    auto compute = [&](auto k_tag) {
        constexpr int K = decltype(k_tag)::value;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int jp = 0; jp < 4; jp += 2) {
                const uint2 pk0 = pack_bf16x4(acc[2 * K + i][jp] + bv[jp] + pv[K & 1][i][jp]);
                const uint2 pk1 = pack_bf16x4(acc[2 * K + i][jp + 1] + bv[jp + 1] + pv[K & 1][i][jp + 1]);
                ln_accumulate(pk0, psum[2 * K + i], psq[2 * K + i]);  // canonical order: column tiles ascending, pairs in order
                ln_accumulate(pk1, psum[2 * K + i], psq[2 * K + i]);
                packed[i][jp >> 1] = pair_to_row16(pk0, pk1);
            }
    };
    auto store = [&](auto k_tag) {
        constexpr int K = decltype(k_tag)::value;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            *(uint4*)(outb + lo_of(2 * K + i) + c0) = packed[i][0];
            *(uint4*)(outb + lo_of(2 * K + i) + c1) = packed[i][1];
        }
    };
    using K0 = std::integral_constant<int, 0>;
    using K1 = std::integral_constant<int, 1>;
    using K2 = std::integral_constant<int, 2>;
    using K3 = std::integral_constant<int, 3>;
    load_pos(K0{});
    load_pos(K1{});
    asm volatile("" ::: "memory");
    between();
    asm volatile("" ::: "memory");
    compute(K0{});
    asm volatile("" ::: "memory");
    store(K0{});
    asm volatile("" ::: "memory");
    compute(K1{});
    load_pos(K2{});  // into the registers quarter 0 has left, BEFORE quarter 1's stores
    asm volatile("" ::: "memory");
    store(K1{});
    asm volatile("" ::: "memory");
    compute(K2{});
    load_pos(K3{});
    asm volatile("" ::: "memory");
    store(K2{});
    asm volatile("" ::: "memory");
    compute(K3{});
    store(K3{});
    if (stats) {
        // as EPI_BIAS_RES_STATS: totals over the four lanes (fq) that share a row, ((fq0 + fq1) + (fq2 + fq3))
        float t[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(psum[k]), __float_as_uint(psq[k]), false, false);
            t[k] = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
        }
        float u[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(t[2 * k]), __float_as_uint(t[2 * k + 1]), false, false);
            u[k] = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
        }
        // lane (fq, fr) holds, for k = 0..3, plane (fq & 1) of patch row mw + (fq >> 1) * 16 + fr + 32 k
        float* plane = g.ln_part + ((int64_t)(fq & 1) * (g.N >> 6) + (nw >> 6)) * g.ln_part_rows;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int m = mw + (fq >> 1) * 16 + fr + 32 * k;
            plane[(int64_t)m + m / VIT_NP + 1] = u[k];
        }
    }
}

template <int EPI>
constexpr bool epi_has_fast_path() {
    return EPI == EPI_BIAS || EPI == EPI_BIAS_GELU || EPI == EPI_BIAS_RES || EPI == EPI_LN_BIAS || EPI == EPI_LN_BIAS_GELU ||
           EPI == EPI_BIAS_RES_STATS;
}
