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
// through registers into a ring of three LDS buffers (global loads of tile t+2 are in flight while tile t is computed; the row
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
constexpr int NBUF = 3;                                                                // key/value tiles resident in LDS
constexpr int QB = 256;                                                                // queries per workgroup
constexpr int NQB = (TV_T + QB - 1) / QB;                                              // 26

typedef __attribute__((ext_vector_type(8))) short s16x8;

__device__ __forceinline__ bool tv_is_pad(int tok_index, int ntile) {
    const int tile = tok_index / TV_TOKP, tok = tok_index - tile * TV_TOKP;
    return tok >= TV_TOK || tile >= ntile;
}

// Schedule (round 2, second cut).  The online softmax works on GRANULES of 64 keys (two per staged tile); every wave
// runs the same pieces per granule -- S^T (10 MFMAs), the running maximum + rescale (vector ALU), exponentials +
// O^T += V^T P^T (12 MFMAs beside ~130 vector instructions) -- and ONE barrier per 128-key tile hands the next staged
// tile over.  With all eight waves in the same order the two waves of a SIMD did matrix work at the same time and
// vector work at the same time (6.7 k cycles per tile = the SUM of the 2.8 k of MFMA issue and the ~4 k of vector issue
// of the pair).  Now waves 4-7 run HALF A GRANULE BEHIND waves 0-3: they end an interval with the scores + softmax of
// the tile's second granule and begin the next one with its P.V (16 registers of packed P carried across the
// barrier), so on every SIMD one wave's exponentials sit beside the other's MFMAs throughout (guide: stagger by wave
// >= 4, not by parity).  V of tile t-1 is still read during interval t, so the ring holds THREE tiles (141 KiB).
// Scores arrive in log2 units: 80^-0.5 * log2(e) is folded into W_q at load time (capi_tilevit.hip), for this kernel and for
// the fast form below alike.  run_if: when given, the launch returns at once unless *run_if != 0 (the exact re-run behind the
// fast form).
__global__ __launch_bounds__(512, 2) void attn_fwd_tiles(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out,
                                                         const int32_t* __restrict__ ntiles, const int* __restrict__ run_if) {
    extern __shared__ __attribute__((aligned(16))) char lds[];  // NBUF x (K[128][176 B] | V[128][192 B])
    if (run_if && *(const volatile int*)run_if == 0) return;  // uniform: every wave of every workgroup takes the same way
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2;  // 0: a tile's three pieces inside one interval; 1: half a tile behind
    const int qb = blockIdx.x % NQB, head = (blockIdx.x / NQB) % TV_H, img = blockIdx.x / (NQB * TV_H);
    const int ntile = ntiles[img];
    const char* base = (const char*)qkv + (size_t)img * TV_T * TV_LD + head * (TV_DH * 2);
    const int r = lane & 31, hh = lane >> 5;
    const int q = qb * QB + wave * 32 + r;  // this lane's query
    const bool wave_active = qb * QB + wave * 32 < TV_T;  // wave-uniform
    const bool q_pad = q >= TV_T || tv_is_pad(q, ntile);
    const bool wave_has_pad_query = __ballot(q_pad) != 0;

    // V columns 80..95 of every row, all buffers: zero once (never overwritten)
    for (int i = tid; i < NBUF * KT * 2; i += 512) {
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

    // staging: 2560 16-byte chunks per tile (128 rows x (10 of K + 10 of V)); four threads share a row and take chunks
    // q4, q4 + 4, ... q4 + 16 of its 20, so one row offset per thread addresses all five (five NAMED registers: an array
    // captured by a lambda lands in scratch memory)
    uint4 stage0, stage1, stage2, stage3, stage4;
    const int srow = tid >> 2, q4 = tid & 3;
    // chunk j of a row: K bytes [16 j, 16 j + 16) for j < 10, V bytes [16 (j - 10), ...) else
    auto g_chunk = [&](int i) { const int j = q4 + 4 * i; return (j < 10 ? TV_D * 2 + j * 16 : 2 * TV_D * 2 + (j - 10) * 16); };
    auto l_chunk = [&](int i) { const int j = q4 + 4 * i; return (j < 10 ? srow * KROW + j * 16 : KT * KROW + srow * VROW + (j - 10) * 16); };
    // rows past the end of the sequence (last tile) re-read the last row: finite data, masked later
#define TV_LOAD(i, t) *(const uint4*)(base + (size_t)min((t) * KT + srow, TV_T - 1) * TV_LD + g_chunk(i))
#define LOAD_TILE(t)                \
    stage0 = TV_LOAD(0, t);         \
    stage1 = TV_LOAD(1, t);         \
    stage2 = TV_LOAD(2, t);         \
    stage3 = TV_LOAD(3, t);         \
    stage4 = TV_LOAD(4, t);
#define STORE_TILE(buf)                                  \
    *(uint4*)(lds + (buf) * BUFB + l_chunk(0)) = stage0;   \
    *(uint4*)(lds + (buf) * BUFB + l_chunk(1)) = stage1;   \
    *(uint4*)(lds + (buf) * BUFB + l_chunk(2)) = stage2;   \
    *(uint4*)(lds + (buf) * BUFB + l_chunk(3)) = stage3;   \
    *(uint4*)(lds + (buf) * BUFB + l_chunk(4)) = stage4;

    constexpr int NT = (TV_T + KT - 1) / KT;  // 51 key tiles (the last holds 32 keys)
    LOAD_TILE(0)
    STORE_TILE(0)
    LOAD_TILE(1)
    __syncthreads();

    constexpr float sc = 1.0f;  // the scale is in W_q (see above); x * 1.0f is exact, the expressions below keep their shape
    float m_run = -INFINITY, l_run = 0.f;                            // running maximum (log2 units) and sum of this lane's query
    f32x16 o[3];
#pragma unroll
    for (int db = 0; db < 3; ++db)
#pragma unroll
        for (int e = 0; e < 16; ++e) o[db][e] = 0.f;

    // transposed-read lane roles (as attention.hip): group g of 16 lanes, lane 4 tq + tp supplies key row tq, dims 4 tp..
    const int g = lane >> 4, li = lane & 15, tq = li >> 2, tp = li & 3;
    const int v_lane_off = (4 * (g >> 1) + tq) * VROW + (16 * (g & 1) + 4 * tp) * 2;

    // state of the 64-key granule whose P.V is still to come (waves 4-7 carry it across the barrier)
    f32x16 s[2];          // s[u][e] = score of key 64 g + 32u + (e&3) + 8(e>>2) + 4hh
    bf16x8 pf[4];         // its probabilities, packed: the B operands of the four P.V steps
    unsigned msk = 0;     // bit 16 u + e: score s[u][e] does not count

    // ---- S^T for the two 32-key sub-tiles of a granule
    auto scores = [&](const char* Kl, int half) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            f32x16 a;
#pragma unroll
            for (int e = 0; e < 16; ++e) a[e] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 5; ++ks) {
                const bf16x8 kf = *(const bf16x8*)(Kl + (half * 64 + u * 32 + r) * KROW + (2 * ks + hh) * 16);
                a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], a, 0, 0, 0);
            }
            s[u] = a;
        }
    };
    // ---- P = exp2(s*sc - ref) into pf, returns this lane's part of the row sum
    auto probabilities = [&](float ref, auto masked_tag) {
        constexpr bool MASKED = decltype(masked_tag)::value;
        float psum = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float p = __builtin_amdgcn_exp2f(fmaf(s[i >> 1][8 * (i & 1) + j], sc, -ref));
                if (MASKED && ((msk >> (16 * (i >> 1) + 8 * (i & 1) + j)) & 1u)) p = 0.f;
                psum += p;
                pf[i][j] = (bf16_t)p;
            }
        return psum;
    };
    // ---- online softmax of one granule, DEFER-MAX form (guide T13).  The probabilities are computed against the
    // reference point the query ALREADY has (m_run), which does not depend on this granule's maximum, so the
    // exponentials and the maximum's dependent chain (tree, cross-half swap, compare) overlap instead of running one
    // after the other; m_run only moves -- with the rescale of O and l and a second pass over the probabilities -- when
    // some query of the wave found a score more than DEFER above its reference (then p <= 2^DEFER = 256: harmless in
    // f32 sums and in bf16 P).  The first granule always takes that path (m_run = -inf).
    // Keys past the sequence end (last tile) never count; (padding query, padding key) pairs are masked.  The mask is a
    // per-lane bit set consulted where a score is USED -- the score registers themselves, fresh MFMA results, are left
    // alone -- and only granules that can contain a masked pair take that form of the body.
    constexpr float DEFER = 8.0f;
    auto softmax = [&](int k0, auto masked_tag) {
        constexpr bool MASKED = decltype(masked_tag)::value;
        msk = 0;
        if constexpr (MASKED) {
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int key = k0 + 32 * u + (e & 3) + 8 * (e >> 2) + 4 * hh;
                    if (key >= TV_T || (q_pad && tv_is_pad(key, ntile))) msk |= 1u << (16 * u + e);
                }
        }
        // a query with no countable key so far has m_run = -inf: 0 stands in as the reference point (its p are masked)
        const float ref_old = m_run == -INFINITY ? 0.f : m_run;
        float psum = probabilities(ref_old, masked_tag);
        // max over the raw scores, scaled once: fl(s * sc) is monotone in s (sc > 0), so this is max_i fl(s_i * sc)
        float raw = -INFINITY;
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int e = 0; e < 16; ++e) raw = fmaxf(raw, (MASKED && ((msk >> (16 * u + e)) & 1u)) ? -INFINITY : s[u][e]);
        float mx = raw * sc;
        {   // the other half of the wave holds the other keys of this query: one v_permlane32_swap instead of an LDS round trip
            const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
            mx = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
        }
        if (__ballot(mx > m_run + DEFER) != 0) {  // wave-uniform (-inf + DEFER = -inf: any countable key moves a fresh query)
            const float m_new = fmaxf(m_run, mx);
            const float ref = m_new == -INFINITY ? 0.f : m_new;
            const float alpha = __builtin_amdgcn_exp2f(m_run - ref);  // 0 when m_run = -inf, 1 for the lanes that did not move
            m_run = m_new;
            l_run *= alpha;
#pragma unroll
            for (int db = 0; db < 3; ++db)
#pragma unroll
                for (int e = 0; e < 16; ++e) o[db][e] *= alpha;
            psum = probabilities(ref, masked_tag);
        }
        l_run += psum;
    };
    // ---- O^T += V^T . P^T in 4 steps of 16 keys
    auto pv = [&](const char* Vl, int half) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int db = 0; db < 3; ++db) {
                const char* va = Vl + (half * 4 + i) * 16 * VROW + v_lane_off + db * 64;
                const s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4*)va);
                const s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4*)(va + 8 * VROW));
                const s16x8 vf = __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7);
                o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, vf), pf[i], o[db], 0, 0, 0);
            }
    };
    auto granule_masked = [&](int k0) {  // wave-uniform
        const int tok0 = k0 % TV_TOKP;
        const bool tail = k0 + 64 > TV_T;
        const bool has_pad_key = tok0 + 64 > TV_TOK || (k0 + 63) / TV_TOKP >= ntile;
        return tail || (wave_has_pad_query && has_pad_key);
    };
    auto front = [&](const char* Kl, int k0, int half) {  // scores + softmax of one granule: leaves its P in pf
        scores(Kl, half);
        if (granule_masked(k0)) softmax(k0, std::true_type{});
        else softmax(k0, std::false_type{});
    };
    auto back = [&](const char* Vl, int half) { pv(Vl, half); };  // P.V of the granule `front` left pending

    int buf = 0, buf_prev = 0;  // ring slots of tiles t and t-1
    for (int t = 0; t < NT; ++t) {
        if (t > 0) __syncthreads();  // tile t is in LDS (written one interval ago); slot (t+1) % 3 was last read in interval t-1
        const int buf_next = buf == NBUF - 1 ? 0 : buf + 1;
        if (t + 1 < NT) {
            STORE_TILE(buf_next)
        }
        if (t + 2 < NT) {
            LOAD_TILE(t + 2)
        }
        if (wave_active) {
            const char* Kl = lds + buf * BUFB;
            const char* Vl = Kl + KT * KROW;
            if (grp == 0) {
                front(Kl, t * KT, 0);
                back(Vl, 0);
                front(Kl, t * KT + 64, 1);
                back(Vl, 1);
            } else {
                if (t > 0) back(lds + buf_prev * BUFB + KT * KROW, 1);
                front(Kl, t * KT, 0);
                back(Vl, 0);
                front(Kl, t * KT + 64, 1);
            }
        }
        buf_prev = buf;
        buf = buf_next;
    }
    if (!wave_active) return;
    if (grp == 1) back(lds + buf_prev * BUFB + KT * KROW, 1);  // the last granule: its V stays in the ring, nothing is staged any more

    l_run += __shfl_xor(l_run, 32, 64);
    const float inv = l_run > 0.f ? __builtin_amdgcn_rcpf(l_run) : 0.f;
    // o[db][4*rg + j] = O[q][32db + 8rg + 4hh + j]; pair the lane halves into 16-byte stores; dims >= 80 do not exist
    // (T % 32 == 0: a wave's 32 queries are all inside the sequence or all outside, so the swaps below see whole waves)
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
    }
}


// ---------------------------------------------------------------------------------------------------------------------
// FAST form (round 4).  The kernel above is bound by its vector work: per score one fma, one exponential, one running
// maximum, one row-sum add and half a conversion (5.5 issue slots against 0.7 slots of matrix work at head dim 80), and a
// wave's matrix and vector phases alternate.  Here a score costs ONE exponential and half a conversion:
//   * reference point inside the matrix pipe: the accumulators of S^T = K . Q^T start from a 16-register block holding
//     -ref of the lane's query (the C operand of a sub-tile's first MFMA; Q arrives in log2 units), so a score leaves the
//     pipe ready for exp2 -- no subtraction;
//   * row sums by the matrix pipe: V is staged with head dims 80..95 as padding (the third 32-row block of O^T is a whole
//     MFMA either way); column 80 holds 1.0 instead of 0, so O^T row 80 IS the sum of the (bf16-rounded) probabilities --
//     no adds, and numerator and denominator see the same rounded values;
//   * NO running maximum: softmax(s) = exp2(s - c) / sum for ANY c; the maximum only buys range.  c starts as the maximum
//     over the query's first 32 keys and is re-centred from the ROW SUM: after every 128-key tile, a query whose sum passed
//     2^60 moves its reference by the sum's exponent (O and the sum scale by an exact power of two, the -ref block moves
//     with it).  So ref >= (largest score so far) - 60 after every tile, and a probability overflows only if a score
//     jumps more than ~67 log2 units (46 nats of q.k / sqrt(d)) above everything the query met before within one tile:
//     then the sum is inf / NaN, the lane raises `guard`, and the exact kernel above -- launched right behind with
//     run_if = guard -- redoes the launch.  Every finite input gets a correct result; ordinary inputs never re-run.
//   * software pipeline per wave over 32-key sub-tiles: the five S^T MFMAs of sub-tile k+1 are issued around the
//     exponentials of sub-tile k (two score blocks alive), then the six P.V MFMAs of sub-tile k with their transposed V
//     reads; no stagger between the wave groups, two LDS buffers, one barrier per 128-key tile.
// (padding query, padding key) pairs are masked as in the exact kernel (probability 0), in sub-tiles that can contain one.
// DBG: ablation switches, instantiated in the diagnostic build only (results invalid): 1 no exponentials, 2 no P.V MFMAs, 4 no
// S^T MFMAs, 8 no barrier, 16 no K/V staging; 32 phase stamps (results valid, timing perturbed).  (As a run-time argument the switches cost the product kernel 15 ms per 8-image
// pass: every wave-uniform branch ends a scheduling region.)
template <int DBG>
__global__ __launch_bounds__(512, 2) void attn_fwd_tiles_fast(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out,
                                                              const int32_t* __restrict__ ntiles, int* __restrict__ guard, float guard_limit) {
    constexpr int dbg = DBG;
    extern __shared__ __attribute__((aligned(16))) char lds[];  // 2 x (K[128][176 B] | V[128][192 B])
    constexpr int FB = 2;                                        // buffers
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int qb = blockIdx.x % NQB, head = (blockIdx.x / NQB) % TV_H, img = blockIdx.x / (NQB * TV_H);
    const int ntile = ntiles[img];
    const char* base = (const char*)qkv + (size_t)img * TV_T * TV_LD + head * (TV_DH * 2);
    const int r = lane & 31, hh = lane >> 5;
    const int q = qb * QB + wave * 32 + r;
    const bool wave_active = qb * QB + wave * 32 < TV_T;  // wave-uniform
    const bool q_pad = q >= TV_T || tv_is_pad(q, ntile);
    const bool wave_has_pad_query = __ballot(q_pad) != 0;

    // V columns 80..95 of every row, both buffers, written once: column 80 = 1.0 (the row-sum column), 81..95 = 0
    for (int i = tid; i < FB * KT * 2; i += 512) {
        const int buf = i / (KT * 2), row = (i >> 1) % KT, c = i & 1;
        *(uint4*)(lds + buf * BUFB + KT * KROW + row * VROW + 160 + c * 16) = make_uint4(c == 0 ? 0x3F80u : 0u, 0, 0, 0);
    }

    bf16x8 qf[5];
    {
        const char* qp = base + (size_t)min(q, TV_T - 1) * TV_LD + hh * 16;
#pragma unroll
        for (int ks = 0; ks < 5; ++ks) qf[ks] = *(const bf16x8*)(qp + ks * 32);
    }

    uint4 stage0, stage1, stage2, stage3, stage4;
    const int srow = tid >> 2, q4 = tid & 3;
    auto g_chunk = [&](int i) { const int j = q4 + 4 * i; return (j < 10 ? TV_D * 2 + j * 16 : 2 * TV_D * 2 + (j - 10) * 16); };
    auto l_chunk = [&](int i) { const int j = q4 + 4 * i; return (j < 10 ? srow * KROW + j * 16 : KT * KROW + srow * VROW + (j - 10) * 16); };

    constexpr int NT = (TV_T + KT - 1) / KT;  // 51 key tiles; the last holds 32 keys = exactly one sub-tile (T % 32 == 0)
    LOAD_TILE(0)
    STORE_TILE(0)
    LOAD_TILE(1)
    __syncthreads();

    f32x16 o[3], negm;
#pragma unroll
    for (int e = 0; e < 16; ++e) o[0][e] = o[1][e] = o[2][e] = negm[e] = 0.f;

    const int g = lane >> 4, li = lane & 15, tq = li >> 2, tp = li & 3;
    const int v_lane_off = (4 * (g >> 1) + tq) * VROW + (16 * (g & 1) + 4 * tp) * 2;
    const int k_lane_off = r * KROW + hh * 16;

    // fragment reads run ONE PHASE AHEAD of the MFMAs that consume them (a phase = the 5 S^T or the 6 P.V MFMAs of a sub-tile,
    // 160-190 cycles of matrix issue: more than an LDS round trip under load), into named register sets; sched_barriers
    // keep hipcc from sinking the reads back in front of their MFMAs (it placed them one or two MFMAs ahead: 35 % of the
    // wave cycles sat in s_waitcnt, profiles/round4_pmc_tilevit_*)
    bf16x8 kf[5];
    s16x4 vlo[6], vhi[6];
    auto k_reads = [&](const char* Kl, int k) {
        const char* kp = Kl + k * 32 * KROW + k_lane_off;
#pragma unroll
        for (int ks = 0; ks < 5; ++ks) kf[ks] = *(const bf16x8*)(kp + ks * 32);
    };
    auto v_reads = [&](const char* Vl, int k) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int db = 0; db < 3; ++db) {
                const char* va = Vl + (k * 2 + i) * 16 * VROW + v_lane_off + db * 64;
                vlo[i * 3 + db] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4*)va);
                vhi[i * 3 + db] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4*)(va + 8 * VROW));
            }
    };
    f32x16 sa, sb;   // score blocks of two consecutive sub-tiles
    bf16x8 pf[2];    // probabilities of the sub-tile whose P.V comes next
#define TV_FENCE __builtin_amdgcn_sched_barrier(0);
    // DBG bit 32: s_memtime stamps (diagnostic build; printed by one workgroup).  Each stamp drains the LDS queue (s_memtime
    // returns through lgkmcnt), so a "reads" segment shows the full round trip of the reads it issued and the segment after it
    // none of that wait: read the SHARES, not the kernel's length.  st_acc: 0 tile barrier, 1 fragment reads + staging chunk,
    // 2 S^T (+ exponentials), 3 P.V, 4 rest (reference, re-centring, loop)
    unsigned long long st_acc[5] = {0, 0, 0, 0, 0}, st_prev = 0;
    auto stamp = [&](int cat) {
        if constexpr ((DBG & 32) != 0) {
            unsigned long long now;
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now)::"memory");
            __builtin_amdgcn_sched_barrier(0);
            st_acc[cat] += now - st_prev;
            st_prev = now;
        }
    };
    // S^T of the sub-tile whose K fragments are in kf: 5 MFMAs, the first one starts from the -ref block
    auto scores = [&](f32x16& dst) {
        if (dbg & 4) {  // ablation: no S^T MFMAs
            asm volatile("" ::"v"(kf[0]), "v"(kf[4]));
            dst = negm;
            return;
        }
        f32x16 a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[0], qf[0], negm, 0, 0, 0);
#pragma unroll
        for (int ks = 1; ks < 5; ++ks) a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[ks], qf[ks], a, 0, 0, 0);
        dst = a;
    };
    // P = exp2(score) (the score is already relative to the reference), packed as the B operands of the two P.V steps;
    // (padding query, padding key) pairs do not count (sub-tiles that can hold one: rare, wave-uniform)
    auto probabilities = [&](const f32x16& sv, int k0) {
        if (dbg & 1) {  // ablation: no exponentials (one multiply instead)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 8; ++j) pf[i][j] = (bf16_t)(sv[8 * i + j] * 0.001f);
            return;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) pf[i][j] = (bf16_t)__builtin_amdgcn_exp2f(sv[8 * i + j]);
        const int tok0 = k0 % TV_TOKP;
        if (wave_has_pad_query && (tok0 + 32 > TV_TOK || (k0 + 31) / TV_TOKP >= ntile)) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int key = k0 + (e & 3) + 8 * (e >> 2) + 4 * hh;
                if (q_pad && tv_is_pad(key, ntile)) pf[e >> 3][e & 7] = (bf16_t)0.f;
            }
        }
    };
    // O^T += V^T . P^T for the 32 keys of the sub-tile whose V fragments are in vlo / vhi: 2 steps of 16 keys x 3 blocks of 32
    // head dims (block 2: dims 64..79, the row-sum column 80, zeros)
    auto pv = [&]() {
        if (dbg & 2) {  // ablation: no P.V MFMAs (the operands stay alive)
            asm volatile("" ::"v"(pf[0]), "v"(pf[1]), "v"(vlo[0]), "v"(vhi[5]));
            return;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int db = 0; db < 3; ++db) {
                const s16x8 vf = __builtin_shufflevector(vlo[i * 3 + db], vhi[i * 3 + db], 0, 1, 2, 3, 4, 5, 6, 7);
                o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, vf), pf[i], o[db], 0, 0, 0);
            }
    };
    // the row sum of this lane's query so far: row 80 of O^T = element 8 of block 2 in the lower half of the wave
    auto row_sum = [&]() {
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(o[2][8]), __float_as_uint(o[2][8]), false, false);
        return __uint_as_float(sw[0]);  // lanes 32..63 receive lanes 0..31's value, lanes 0..31 keep their own
    };

    // range: re-centre the queries whose row sum passed 2^60 (rare; exact powers of two)
    auto recentre = [&]() {
        const float l = row_sum();
        if (__ballot(l > 0x1p60f) != 0) {
            const int ex = l > 0x1p60f ? __builtin_amdgcn_frexp_expf(l) : 0;  // l = f * 2^ex, f in [0.5, 1)
            const float down = __builtin_amdgcn_ldexpf(1.0f, -ex);           // inf / NaN sums: guard at the end
#pragma unroll
            for (int db = 0; db < 3; ++db)
#pragma unroll
                for (int e = 0; e < 16; ++e) o[db][e] *= down;
#pragma unroll
            for (int e = 0; e < 16; ++e) negm[e] -= (float)ex;
        }
    };
    // Waves 4..7 run ONE PHASE behind waves 0..3: they keep the P.V of a tile's last sub-tile (its probabilities and its V
    // fragments, read before the barrier: 32 registers that exist anyway) for the start of the next interval.  The phases of
    // a wave alternate "S^T of the next sub-tile beside the exponentials of this one" (MFMAs 64 cycles apart, the vector
    // port busy) and "P.V" (six MFMAs back to back, no vector work), so with the shift one wave of a SIMD is in the first
    // kind while its partner is in the second -- in lockstep both did the same thing at the same time and the matrix pipe
    // saw the SUM of the two (52 % busy).  (Guide: split the roles of SIMD partners by wave >= 4, not by parity.)
    const int grp = wave >> 2;
    bool carried = false;  // grp 1: a P.V is pending
    int buf = 0;
    if constexpr ((DBG & 32) != 0) st_prev = __builtin_amdgcn_s_memtime();
    for (int t = 0; t < NT; ++t) {
        stamp(4);
        if (t > 0 && !(dbg & 8)) __syncthreads();
        stamp(0);  // tile t is in LDS (written one interval ago); the other buffer's readers are done
        // K/V staging, spread over the interval: the five 16-byte chunks a thread moves per tile go to LDS (tile t + 1, loaded
        // one interval ago) and are re-requested (tile t + 2) one at a time between the phases below -- all 2560 stores of a
        // tile right behind the barrier kept the LDS write path (~77 B/clk) busy for ~500 cycles in which no wave computed
#define TV_STAGE(i)                                                                      \
        if (!(dbg & 16)) {                                                                   \
            if (t + 1 < NT) *(uint4*)(lds + (buf ^ 1) * BUFB + l_chunk(i)) = stage##i;       \
            if (t + 2 < NT) stage##i = TV_LOAD(i, t + 2);                                    \
        }
        if (!wave_active) {
            TV_STAGE(0) TV_STAGE(1) TV_STAGE(2) TV_STAGE(3) TV_STAGE(4)
        }
        if (wave_active) {
            const char* Kl = lds + buf * BUFB;
            const char* Vl = Kl + KT * KROW;
            const int nsub = t == NT - 1 ? (TV_T - t * KT) / 32 : 4;  // wave-uniform; 1 in the last tile
            k_reads(Kl, 0);
            TV_FENCE
            TV_STAGE(0)
            TV_FENCE
            stamp(1);
            if (carried) {  // grp 1: the last sub-tile of the previous tile
                pv();
                recentre();
                TV_FENCE
                stamp(3);
            }
            scores(sa);  // t = 0: negm = 0, raw scores
            TV_FENCE
            stamp(2);
            k_reads(Kl, 1);  // rows past the sequence end (last tile) hold finite data (TV_LOAD clamps): unused
            v_reads(Vl, 0);
            TV_FENCE
            stamp(1);
            if (t == 0) {
                // reference point of every query: the maximum over its first 32 keys (tokens 0..31 of tile 0: never padding)
                float mx = fmaxf(fmaxf(sa[0], sa[1]), sa[2]);
#pragma unroll
                for (int e = 3; e < 15; e += 2) mx = fmaxf(fmaxf(mx, sa[e]), sa[e + 1]);
                mx = fmaxf(mx, sa[15]);
                const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
                mx = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    negm[e] = -mx;
                    sa[e] -= mx;  // the one sub-tile whose scores were formed before the reference existed
                }
            }
            if (nsub > 1) {
                scores(sb);  // sub-tile 1 beside the exponentials of sub-tile 0
                probabilities(sa, t * KT);
                TV_FENCE
                stamp(2);
                k_reads(Kl, 2);
                TV_STAGE(1)
                TV_FENCE
                stamp(1);
                pv();
                TV_FENCE
                stamp(3);
                v_reads(Vl, 1);
                TV_STAGE(2)
                TV_FENCE
                stamp(1);
                scores(sa);  // sub-tile 2
                probabilities(sb, t * KT + 32);
                TV_FENCE
                stamp(2);
                k_reads(Kl, 3);
                TV_STAGE(3)
                TV_FENCE
                stamp(1);
                pv();
                TV_FENCE
                stamp(3);
                v_reads(Vl, 2);
                TV_STAGE(4)
                TV_FENCE
                stamp(1);
                scores(sb);  // sub-tile 3
                probabilities(sa, t * KT + 64);
                TV_FENCE
                stamp(2);
                pv();
                TV_FENCE
                stamp(3);
                v_reads(Vl, 3);
                probabilities(sb, t * KT + 96);
                TV_FENCE
                stamp(1);
            } else {
                probabilities(sa, t * KT);
                TV_FENCE
                TV_STAGE(1) TV_STAGE(2) TV_STAGE(3) TV_STAGE(4)  // the last tile: nothing is staged any more (t + 1 = NT)
            }
            if (grp == 0) {
                pv();
                recentre();
                stamp(3);
            } else {
                carried = true;  // after the barrier (the fragment reads complete in front of it)
            }
        }
        buf ^= 1;
    }
    if constexpr ((DBG & 32) != 0) {
        stamp(4);
        if (blockIdx.x == 300 && lane == 0 && (wave == 0 || wave == 4))
            printf("tattn stamps wave %d (cycles per 128-key tile): barrier %llu | reads+staging %llu | S^T+exp %llu | P.V %llu | rest %llu\n", wave,
                   st_acc[0] / NT, st_acc[1] / NT, st_acc[2] / NT, st_acc[3] / NT, st_acc[4] / NT);
    }
    if (carried) pv();
    if (!wave_active) return;

    const float l = row_sum();
    // !(l < limit) also catches inf and NaN; the sum is >= ~0.5 by construction (the reference key contributes 1)
    if (guard && !(l < guard_limit) && q < TV_T) *guard = 1;  // guard_limit = 2^100 (0.25 in the forced-re-run test mode)
    const float inv = __builtin_amdgcn_rcpf(l);
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
    }
}

#undef TV_FENCE
#undef TV_STAGE
}  // namespace

// qkv [n * 6432, 3840] bf16 -> out [n * 6432, 1280] bf16; ntiles_dev int32[n] (tiles the image uses, 1..4).
// guard != nullptr: the fast form, then the exact kernel with run_if = guard (returns at once unless a row of the fast
// launch left its range; *guard is zeroed by the caller per pass); force_redo: the fast form raises the guard always (test).
hipError_t launch_attention_tiles(const void* qkv, void* out, const int32_t* ntiles_dev, int n, hipStream_t s, int* guard, bool force_redo) {
    if (n <= 0) return hipSuccess;
    static_assert(TV_T % 32 == 0, "a wave's queries are all real rows or none; the last key tile ends on a sub-tile boundary");
    if (hipError_t e = ensure_dynamic_lds((const void*)attn_fwd_tiles, NBUF * BUFB); e != hipSuccess) return e;
    if (guard) {
#define TV_LAUNCH_FAST(D)                                                                                                            \
    {                                                                                                                                \
        if (hipError_t e = ensure_dynamic_lds((const void*)attn_fwd_tiles_fast<D>, 2 * BUFB); e != hipSuccess) return e;              \
        hipLaunchKernelGGL(attn_fwd_tiles_fast<D>, dim3(n * TV_H * NQB), dim3(512), 2 * BUFB, s, (const bf16_t*)qkv, (bf16_t*)out,  \
                           ntiles_dev, guard, force_redo ? 0.25f : 0x1p100f);                                                         \
    }
#ifdef MME_DIAG
        switch (diag_env("MME_TATTN_DEBUG") ? atoi(diag_env("MME_TATTN_DEBUG")) : 0) {
            case 1: TV_LAUNCH_FAST(1) break;
            case 2: TV_LAUNCH_FAST(2) break;
            case 4: TV_LAUNCH_FAST(4) break;
            case 6: TV_LAUNCH_FAST(6) break;
            case 7: TV_LAUNCH_FAST(7) break;
            case 8: TV_LAUNCH_FAST(8) break;
            case 16: TV_LAUNCH_FAST(16) break;
            case 24: TV_LAUNCH_FAST(24) break;
            case 31: TV_LAUNCH_FAST(31) break;
            case 32: TV_LAUNCH_FAST(32) break;
            default: TV_LAUNCH_FAST(0)
        }
#else
        TV_LAUNCH_FAST(0)
#endif
#undef TV_LAUNCH_FAST
        if (hipError_t e = hipGetLastError(); e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(attn_fwd_tiles, dim3(n * TV_H * NQB), dim3(512), NBUF * BUFB, s, (const bf16_t*)qkv, (bf16_t*)out, ntiles_dev, (const int*)guard);
    return hipGetLastError();
}
