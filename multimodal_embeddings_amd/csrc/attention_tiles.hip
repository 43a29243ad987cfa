// Tile-ViT attention: the Mllama vision tower's self-attention (transformers modeling_mllama.py,
// MllamaVisionAttention + eager_attention_forward + _prepare_aspect_ratio_attention_mask).
//
// One image = ONE sequence of T = 4 tiles x 1608 tokens = 6432 (1601 real tokens per tile, padded to a multiple of 8),
// 16 heads of 80, softmax(Q K^T / sqrt(80) + mask) V with the softmax in f32.  The mask is additive finfo.min where
// BOTH the query and the key are padding (a token >= 1601 of its tile, or any token of a tile the image does not
// use); every other pair attends -- valid queries see padding keys, as the reference's model does.
//
// Flash-style, no score matrix: a workgroup owns 256 queries of one (image, head) -- 8 waves x 32 queries, the Q
// fragments stay in registers -- and walks the keys in tiles of 128 with an online softmax (running maximum and sum per
// query, both in-lane: S^T = K . Q^T puts the query on the lane and the keys in the accumulator registers, and the
// exponentiated tile is already the B operand of O^T = V^T . P^T, as in attention.hip).  K and V tiles are staged
// through registers into two LDS buffers (global loads of tile t+2 are in flight while tile t is computed; the row
// pitches -- K 176 B, V 192 B -- make the b128 fragment reads and the transposed b64 reads conflict-free; V is padded
// to 96 head dims with zeros so that the third 32-row block of O^T is a whole MFMA).
#include <type_traits>

#include "common.h"
#include "kernels.h"

namespace {

constexpr int TV_D = 1280, TV_H = 16, TV_DH = 80;
constexpr int TV_TOK = 1601, TV_TOKP = 1608, TV_TILES = 4, TV_T = TV_TILES * TV_TOKP;  // 6432
constexpr int TV_LD = 3 * TV_D * 2;                                                    // byte pitch of a fused QKV row
constexpr int KT = 128;                                                                // keys per tile
constexpr int KROW = 176, VROW = 192;                                                  // LDS row pitches
constexpr int BUFB = KT * (KROW + VROW);                                               // 47 104 B per buffer
constexpr int QB = 256;                                                                // queries per workgroup
constexpr int NQB = (TV_T + QB - 1) / QB;                                              // 26

typedef __attribute__((ext_vector_type(8))) short s16x8;

#define S_BARRIER() asm volatile("s_barrier" ::: "memory")

__device__ __forceinline__ bool tv_is_pad(int tok_index, int ntile) {
    const int tile = tok_index / TV_TOKP, tok = tok_index - tile * TV_TOKP;
    return tok >= TV_TOK || tile >= ntile;
}

__global__ __launch_bounds__(512, 2) void attn_fwd_tiles(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out,
                                                         const int32_t* __restrict__ ntiles) {
    extern __shared__ __attribute__((aligned(16))) char lds[];  // 2 x (K[128][176 B] | V[128][192 B])
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int qb = blockIdx.x % NQB, head = (blockIdx.x / NQB) % TV_H, img = blockIdx.x / (NQB * TV_H);
    const int ntile = ntiles[img];
    const char* base = (const char*)qkv + (size_t)img * TV_T * TV_LD + head * (TV_DH * 2);
    const int r = lane & 31, hh = lane >> 5;
    const int q = qb * QB + wave * 32 + r;  // this lane's query
    const bool wave_active = qb * QB + wave * 32 < TV_T;  // wave-uniform
    const bool q_pad = q >= TV_T || tv_is_pad(q, ntile);
    const bool wave_has_pad_query = __ballot(q_pad) != 0;

    // V columns 80..95 of every row, both buffers: zero once (never overwritten)
    for (int i = tid; i < 2 * KT * 2; i += 512) {
        const int buf = i / (KT * 2), row = (i >> 1) % KT, c = i & 1;
        *(uint4*)(lds + buf * BUFB + KT * KROW + row * VROW + 160 + c * 16) = make_uint4(0, 0, 0, 0);
    }

    // Q fragments: B operand of S^T = K . Q^T; element j of k-step ks is Q[q][16 ks + 8 hh + j]
    bf16x8 qf[5];
    {
        const char* qp = base + (size_t)min(q, TV_T - 1) * TV_LD + hh * 16;
#pragma unroll
        for (int ks = 0; ks < 5; ++ks) qf[ks] = *(const bf16x8*)(qp + ks * 32);
    }

    // staging: 2560 16-byte chunks per tile (K: 128 rows x 10, then V), five per thread, in five NAMED registers
    // (an array captured by a lambda lands in scratch memory)
    uint4 stage0, stage1, stage2, stage3, stage4;
    int g_off[5], l_off[5];  // per-chunk global offset (without the key tile) and LDS offset: loop invariants
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const int c = tid + 512 * i, isv = c >= 1280, cc = isv ? c - 1280 : c;
        const int row = cc / 10, ch = cc - row * 10;
        g_off[i] = row * TV_LD + (isv ? 2 : 1) * (TV_D * 2) + ch * 16;
        l_off[i] = (isv ? KT * KROW + row * VROW : row * KROW) + ch * 16;
    }
    // rows past the end of the sequence (last tile) re-read the last row: finite data, masked later
#define TV_LOAD(i, t) *(const uint4*)(base + (size_t)(t) * KT * TV_LD + min(g_off[i], (TV_T - 1 - (t) * KT) * TV_LD + g_off[i] % TV_LD))
#define LOAD_TILE(t)                \
    stage0 = TV_LOAD(0, t);         \
    stage1 = TV_LOAD(1, t);         \
    stage2 = TV_LOAD(2, t);         \
    stage3 = TV_LOAD(3, t);         \
    stage4 = TV_LOAD(4, t);
#define STORE_TILE(buf)                                  \
    *(uint4*)(lds + (buf) * BUFB + l_off[0]) = stage0;   \
    *(uint4*)(lds + (buf) * BUFB + l_off[1]) = stage1;   \
    *(uint4*)(lds + (buf) * BUFB + l_off[2]) = stage2;   \
    *(uint4*)(lds + (buf) * BUFB + l_off[3]) = stage3;   \
    *(uint4*)(lds + (buf) * BUFB + l_off[4]) = stage4;

    constexpr int NT = (TV_T + KT - 1) / KT;  // 51 key tiles (the last holds 32 keys)
    LOAD_TILE(0)
    STORE_TILE(0)
    LOAD_TILE(1)
    __syncthreads();

    const float sc = 0.11180339887498949f * 1.44269504088896341f;  // 80^-0.5 * log2(e)
    float m_run = -INFINITY, l_run = 0.f;                            // running maximum (log2 units) and sum of this lane's query
    f32x16 o[3];
#pragma unroll
    for (int db = 0; db < 3; ++db)
#pragma unroll
        for (int e = 0; e < 16; ++e) o[db][e] = 0.f;

    // transposed-read lane roles (as attention.hip): group g of 16 lanes, lane 4 tq + tp supplies key row tq, dims 4 tp..
    const int g = lane >> 4, li = lane & 15, tq = li >> 2, tp = li & 3;
    const int v_lane_off = (4 * (g >> 1) + tq) * VROW + (16 * (g & 1) + 4 * tp) * 2;

    for (int t = 0; t < NT; ++t) {
        const int buf = t & 1;
        const char* Kl = lds + buf * BUFB;
        const char* Vl = Kl + KT * KROW;
        if (t > 0) __syncthreads();  // tile t is in LDS (written one iteration ago); everybody left tile t-1
        if (t + 1 < NT) {
            STORE_TILE(buf ^ 1)
        }
        if (t + 2 < NT) {
            LOAD_TILE(t + 2)
        }
        if (!wave_active) continue;

        // ---- S^T for the four 32-key sub-tiles: s[u][e] = score of key t*128 + 32u + (e&3) + 8(e>>2) + 4hh
        f32x16 s[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            f32x16 a;
#pragma unroll
            for (int e = 0; e < 16; ++e) a[e] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 5; ++ks) {
                const bf16x8 kf = *(const bf16x8*)(Kl + (u * 32 + r) * KROW + (2 * ks + hh) * 16);
                a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], a, 0, 0, 0);
            }
            s[u] = a;
        }
        // keys past the sequence end (last tile) never count; (padding query, padding key) pairs are masked.  The mask is a
        // per-lane bit set consulted where a score is USED (maximum, exponential) -- the score registers themselves, fresh
        // MFMA results, are left alone -- and only tiles that can contain a masked pair take that form of the body.
        const int k0 = t * KT;
        const bool tail = k0 + KT > TV_T;
        const int tok0 = k0 % TV_TOKP;
        const bool tile_has_pad_key = tok0 + KT > TV_TOK || (k0 + KT - 1) / TV_TOKP >= ntile;  // wave-uniform
        auto softmax_pv = [&](auto masked_tag) {
            constexpr bool MASKED = decltype(masked_tag)::value;
            unsigned long long msk = 0;  // bit 16 u + e: score s[u][e] does not count
            if constexpr (MASKED) {
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int key = k0 + 32 * u + (e & 3) + 8 * (e >> 2) + 4 * hh;
                        if (key >= TV_T || (q_pad && tv_is_pad(key, ntile))) msk |= 1ull << (16 * u + e);
                    }
            }
            auto dead = [&](int u, int e) { return MASKED && ((msk >> (16 * u + e)) & 1ull) != 0; };
            // ---- online softmax
            float mx = m_run;
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int e = 0; e < 16; ++e) mx = fmaxf(mx, dead(u, e) ? -INFINITY : s[u][e] * sc);
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            // a query whose keys are all masked so far keeps mx = -inf: use 0 as the reference point (all p = 0)
            const float mref = mx == -INFINITY ? 0.f : mx;
            const float alpha = __builtin_amdgcn_exp2f(m_run - mref);  // 0 when m_run = -inf
            m_run = mx;
            l_run *= alpha;
#pragma unroll
            for (int db = 0; db < 3; ++db)
#pragma unroll
                for (int e = 0; e < 16; ++e) o[db][e] *= alpha;
            // ---- P = exp2(s*sc - m), row sums, O^T += V^T . P^T in 8 steps of 16 keys
            float psum = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                bf16x8 pf;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    float pv = __builtin_amdgcn_exp2f(fmaf(s[i >> 1][8 * (i & 1) + j], sc, -mref));
                    if (dead(i >> 1, 8 * (i & 1) + j)) pv = 0.f;
                    psum += pv;
                    pf[j] = (bf16_t)pv;
                }
#pragma unroll
                for (int db = 0; db < 3; ++db) {
                    const char* va = Vl + i * 16 * VROW + v_lane_off + db * 64;
                    const s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4*)va);
                    const s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4*)(va + 8 * VROW));
                    const s16x8 vf = __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7);
                    o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, vf), pf, o[db], 0, 0, 0);
                }
            }
            l_run += psum;
        };
        if (tail || (wave_has_pad_query && tile_has_pad_key)) {
            softmax_pv(std::true_type{});
        } else {
            softmax_pv(std::false_type{});
        }
    }

    if (!wave_active) return;
    l_run += __shfl_xor(l_run, 32, 64);
    const float inv = l_run > 0.f ? __builtin_amdgcn_rcpf(l_run) : 0.f;
    // o[db][4*rg + j] = O[q][32db + 8rg + 4hh + j]; pair the lane halves into 16-byte stores; dims >= 80 do not exist
    if (q < TV_T) {
        bf16_t* op = out + ((size_t)img * TV_T + q) * TV_D + head * TV_DH;
#pragma unroll
        for (int db = 0; db < 3; ++db)
#pragma unroll
            for (int rp = 0; rp < 4; rp += 2) {
                if (db == 2 && rp == 2) continue;
                bf16x4 t0, t1;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    t0[j] = (bf16_t)(o[db][rp * 4 + j] * inv);
                    t1[j] = (bf16_t)(o[db][(rp + 1) * 4 + j] * inv);
                }
                const uint2 u0 = __builtin_bit_cast(uint2, t0), u1 = __builtin_bit_cast(uint2, t1);
                const auto ax = __builtin_amdgcn_permlane32_swap(u0.x, u1.x, false, false);
                const auto ay = __builtin_amdgcn_permlane32_swap(u0.y, u1.y, false, false);
                *(uint4*)(op + db * 32 + (rp + hh) * 8) = make_uint4(ax[0], ay[0], ax[1], ay[1]);
            }
    } else {
        // lanes without a query still take part in the swaps above?  No: the swaps sit inside `if (q < TV_T)`, so
        // a wave must be uniform here -- it is: T % 32 == 0, a wave's 32 queries are all inside or all outside.
    }
}

}  // namespace

// qkv [n * 6432, 3840] bf16 -> out [n * 6432, 1280] bf16; ntiles_dev int32[n] (tiles the image uses, 1..4)
hipError_t launch_attention_tiles(const void* qkv, void* out, const int32_t* ntiles_dev, int n, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    static_assert(TV_T % 32 == 0, "a wave's queries are all real rows or none");
    if (hipError_t e = ensure_dynamic_lds((const void*)attn_fwd_tiles, 2 * BUFB); e != hipSuccess) return e;
    hipLaunchKernelGGL(attn_fwd_tiles, dim3(n * TV_H * NQB), dim3(512), 2 * BUFB, s, (const bf16_t*)qkv, (bf16_t*)out, ntiles_dev);
    return hipGetLastError();
}
