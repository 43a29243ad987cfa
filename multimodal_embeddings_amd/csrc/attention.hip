// K5: fused multi-head self-attention for ViT-B/16 (T = 197 tokens, 12 heads, dh = 64).
//
// Restates transformers models/vit/modeling_vit.py:164-189 (softmax(Q K^T / 8) V, softmax in
// f32).  T is short, so there is no online softmax: the whole 32 x 224 score strip of a query
// block lives in accumulator registers.
//
// One 8-wave workgroup walks the 12 heads of ONE crop:
//   * K and V of head h+1 stream into the second LDS buffer by LDS-DMA (global_load_lds, no VGPR
//     round trip) while head h is computed, and the Q fragments of head h+1 are prefetched into
//     registers: after the first head no memory latency is exposed.  One barrier per head.
//   * waves 0..6 own the 7 query blocks of 32 (197 -> 224); wave 7 only feeds DMA.
//   * S^T = K . Q^T with mfma_f32_32x32x16_bf16: the accumulator has the QUERY on the lane and
//     the 32 keys of a tile in its 16 registers (x2 lane halves), so softmax max / sum are in-lane
//     reductions plus one cross-half shuffle, and the exponentiated tile is already the B operand
//     of O^T = V^T . P^T (guide §3, "an accumulator tile as the next MFMA's operand").
//     Normalisation by 1/sum is deferred to the 32 output values per lane.
//   * V^T fragments come from the row-major V image through ds_read_b64_tr_b16.  LDS images:
//     K rows of 128 B with chunk ^= (row>>1)&7 (conflict-free ds_read_b128), V rows of 128 B with
//     the two 64-byte halves swapped when bit 1 of the row is set (conflict-free transposed
//     reads); both swizzles are applied to the DMA source address.
//   * O^T leaves 4 consecutive head-dims per lane; v_permlane32_swap pairs the two lane halves
//     into 16-byte stores.
#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "kernels.h"

namespace {

constexpr int ROWB = VIT_DH * 2;       // 128-byte K/V rows in LDS
constexpr int QKV_LD = 3 * VIT_D * 2;  // 4608-byte rows of the fused QKV activation
constexpr int NPIECE = 25;             // 25 x 8 rows = 200 >= 197
// NB = LDS buffers (heads in flight + 1).  Two buffers of 224 rows (7 key tiles of 32), or three of 208 rows
// (13 P.V steps of 16 keys; the last S^T tile then reads 16 rows past its K image, into the V image of the same
// buffer: finite bf16 bit patterns or not, those scores belong to padded keys and are replaced, never used).
template <int NB> struct AttnGeom {
    static constexpr int TROWS = NB == 3 ? 208 : 224;
    static constexpr int KV_BYTES = TROWS * ROWB;
    static constexpr int BUF_BYTES = 2 * KV_BYTES;
    static constexpr int LDS_BYTES = NB * BUF_BYTES;
};

typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(2))) float f32x2;

#define S_BARRIER() asm volatile("s_barrier" ::: "memory")

// the other half of the wave (lane ^ 32) holds the other keys of this lane's query: exchange by ONE v_permlane32_swap (vector
// ALU) instead of __shfl_xor's ds_bpermute round trip through the LDS, whose latency sits on the head's critical path
__device__ __forceinline__ float other_half(float x) {
    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    // after the swap sw[0] = {lo, lo}, sw[1] = {hi, hi}: the value this lane did not have is the one that differs
    const float a = __uint_as_float(sw[0]), b = __uint_as_float(sw[1]);
    return (threadIdx.x & 32) ? a : b;
}


// STAMP: diagnostic build -- every wave accumulates s_memtime intervals between seven points of a head
// iteration (wait, barrier, request issue, S^T, softmax, P.V, stores) and writes 8 words per wave to `stamps`
// (layout: mme.h, mme_attention_stamps).  No stamp executes in the product kernel.
// PIPE: the softmax rides inside the MFMA loops (default); false = the phase-by-phase form it replaced (A/B).
template <int NB, bool STAMP = false, bool PIPE = true>
__global__ __launch_bounds__(512, 2) void attn_fwd_t197(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out, int B, int hsplit,
                                                         unsigned long long* stamps = nullptr, int dbg_arg = 0) {
    // ablation switches of the STAMPED build only (results invalid): 1 no K re-reads, 2 no K/V requests after the
    // first head, 4 no maximum, 8 no exponentials
    const int dbg = STAMP ? dbg_arg : 0;
    constexpr int TROWS = AttnGeom<NB>::TROWS, KV_BYTES = AttnGeom<NB>::KV_BYTES, BUF_BYTES = AttnGeom<NB>::BUF_BYTES;
    extern __shared__ __attribute__((aligned(16))) char lds[];  // NB x (K | V)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // hsplit workgroups share a crop (small batches: one workgroup per crop would leave most CUs idle and
    // serialise 12 heads on one); this one walks heads [h_begin, h_end)
    const int b = blockIdx.x / hsplit, hpw = VIT_H / hsplit;
    const int h_begin = (blockIdx.x - b * hsplit) * hpw, h_end = h_begin + hpw;
    const char* base = (const char*)qkv + (size_t)b * VIT_T * QKV_LD;

    // V rows 200.. are never written by DMA: zero them once in every buffer (P is 0 there, but
    // 0 * garbage could be NaN).  Rows 197..199 receive clamped copies of row 196 (finite).
    constexpr int ZROWS = TROWS - 200;
    for (int i = tid; i < NB * ZROWS * 8; i += 512) {
        const int buf = i / (ZROWS * 8), r = (i >> 3) % ZROWS, c = i & 7;
        *(uint4*)(lds + buf * BUF_BYTES + KV_BYTES + (200 + r) * ROWB + c * 16) = make_uint4(0, 0, 0, 0);
    }

    // DMA of one head: 25 K pieces + 25 V pieces (8 rows x 128 B each), all issued by wave 7: an
    // LDS-DMA piece stalls its issuer for ~100+ cycles, which the seven computing waves cannot afford
    auto dma_head = [&](int h, int buf) {
        const char* hb = base + h * ROWB;
        char* kdst = lds + buf * BUF_BYTES;
        if (wave != 7) return;  // the wave without a query block does all the staging
        if ((dbg & 2) && h != h_begin) return;
        for (int p = 0; p < 2 * NPIECE; ++p) {
            const bool isv = p >= NPIECE;
            const int pp = isv ? p - NPIECE : p;
            const int row = pp * 8 + (lane >> 3);
            const int slot = lane & 7;
            const int chunk = isv ? (slot ^ (((row >> 1) & 1) << 2)) : (slot ^ ((row >> 1) & 7));
            const char* src = hb + (size_t)min(row, VIT_T - 1) * QKV_LD + (isv ? 2 : 1) * VIT_D * 2 + chunk * 16;
            // LDS-DMA through inline asm, so that hipcc does not know these loads write LDS: told through the builtin it
            // orders every later LDS read behind them with `s_waitcnt vmcnt(0)` -- in the middle of the head iteration,
            // where that also waits for the Q prefetch and the previous head's stores.  The ordering that is needed
            // (pieces landed before the NEXT head reads them) is the counted wait + barrier at the top of the loop.
            const unsigned dst = (unsigned)(size_t)(LDS_AS char*)(kdst + (isv ? KV_BYTES : 0) + pp * 1024);
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep)
                         : "v"(src), "s"(__builtin_amdgcn_readfirstlane(dst))
                         : "memory");
        }
    };

    const int r = lane & 31, hh = lane >> 5;
    const int ksw = (r >> 1) & 7;
    // transposed-read lane roles: group g of 16 lanes, lane 4q+p supplies row q, cols 4p..4p+3
    const int g = lane >> 4, li = lane & 15, tq = li >> 2, tp = li & 3;
    const int vflag = (tq >> 1) & 1;  // bit 1 of the row this lane addresses: selects the swapped half
    const int v_row_off = (4 * (g >> 1) + tq) * ROWB + (16 * (g & 1) + 4 * tp) * 2;
    const int v_off0 = v_row_off + ((0 ^ vflag) << 6), v_off1 = v_row_off + ((1 ^ vflag) << 6);
    const float sc = 0.125f * 1.44269504088896341f;  // dh^-0.5 * log2(e)
    const int q = wave * 32 + r;                      // this lane's query (waves 0..6)
    const bool active = wave < 7;
    const char* qp = base + (size_t)min(q, VIT_T - 1) * QKV_LD + hh * 16;

    bf16x8 qf[4], qn[4];
    dma_head(h_begin, 0);
    if (NB == 3 && h_begin + 1 < h_end) dma_head(h_begin + 1, 1);  // three buffers: K/V run TWO heads ahead
    if (active) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) qf[ks] = *(const bf16x8*)(qp + h_begin * ROWB + ks * 32);
        // Complete these loads HERE.  Left pending into the loop they make hipcc put `s_waitcnt vmcnt(0)` in front of
        // the first MFMA of every head iteration (it cannot tell the iterations apart), which also waits for the Q
        // prefetch of the next head issued a few instructions earlier and for the previous head's output stores:
        // a full memory latency exposed per head (stamped: ~2 000 of the ~10 000 cycles of a head iteration).
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) asm volatile("" : "+v"(qf[ks]));
    }

    unsigned long long st[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_prev = 0;
#define ATTN_STAMP(i)                                                \
    if (STAMP) {                                                     \
        const unsigned long long now = __builtin_amdgcn_s_memtime(); \
        st[i] += now - st_prev;                                      \
        st_prev = now;                                               \
    }
    if (STAMP) st_prev = __builtin_amdgcn_s_memtime();
    const unsigned long long st_c0 = STAMP ? st_prev : 0, st_r0 = STAMP ? __builtin_amdgcn_s_memrealtime() : 0;
    if (wave >= 4) __builtin_amdgcn_s_setprio(1);  // the later-dispatched half loses VALU arbitration otherwise
    for (int h = h_begin; h < h_end; ++h) {
        const int buf = NB == 3 ? (h - h_begin) % 3 : (h - h_begin) & 1;
        const char* Kl = lds + buf * BUF_BYTES;
        const char* Vl = Kl + KV_BYTES;
        // head h has landed (each wave waits for its own pieces), everybody is done with head h-1
        // (the 4 output stores of head h-1 are this wave's youngest vector-memory operations and may
        // stay in flight: vmcnt retires in order and counts stores)
        if (!active) {
            // the staging wave: head h has landed; with three buffers the 50 pieces of head h+1 stay in flight
            if (NB == 3 && h + 1 < h_end) {
                asm volatile("s_waitcnt vmcnt(50) lgkmcnt(0)" ::: "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            }
        } else if (h == h_begin) {
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
        }
        ATTN_STAMP(0)  // own requests landed
        S_BARRIER();
        ATTN_STAMP(1)  // everybody's
        if (NB == 3 && h + 2 < h_end) dma_head(h + 2, (h + 2 - h_begin) % 3);
        if (h + 1 < h_end) {
            if (NB == 2) dma_head(h + 1, buf ^ 1);
            if (active) {
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) qn[ks] = *(const bf16x8*)(qp + (h + 1) * ROWB + ks * 32);
            }
        }
        ATTN_STAMP(2)  // next head's requests issued
        if (active) {
            f32x16 s[7];
            float inv;
            f32x16 o[2];
            s16x8 vf[2][2];
            auto read_v = [&](int i, s16x8 (&dst)[2]) {
#pragma unroll
                for (int db = 0; db < 2; ++db) {
                    const char* va = Vl + i * 16 * ROWB + (db ? v_off1 : v_off0);
                    const s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4*)va);
                    const s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4*)(va + 8 * ROWB));
                    dst[db] = __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7);
                }
            };
            if constexpr (PIPE) {
                // The kernel is bound by vector-ALU and matrix issue per SIMD, not by HBM (stamped build: a wave's
                // S^T, softmax and P.V phases ran back to back and its SIMD partner's did not overlap them).  Here
                // the softmax rides inside the two MFMA loops of the SAME wave: the running maximum of tile kt - 1
                // is taken while the four MFMAs of tile kt execute, and the exponentials / bf16 conversion of
                // P.V step i + 1 are computed between the two MFMAs of step i.
                // S^T key tiles in pairs: the two accumulation chains of a pair alternate in the matrix pipe, so no MFMA
                // waits for its predecessor's result; the K fragments of the next pair are requested first
                bf16x8 kf[2][2][4];  // [pair parity][tile of the pair][k step]
                auto read_k = [&](int kt, bf16x8 (&dst)[4]) {
                    if ((dbg & 1) && kt >= 2) return;
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) dst[ks] = *(const bf16x8*)(Kl + (kt * 32 + r) * ROWB + (((2 * ks + hh) ^ ksw) << 4));
                };
                read_k(0, kf[0][0]);
                read_k(1, kf[0][1]);
                float mx = -INFINITY, mx1 = -INFINITY;
                auto tile_max = [&](const f32x16& t) {
                    if (dbg & 4) return;
#pragma unroll
                    for (int e = 0; e < 16; e += 4) {
                        mx = fmaxf(fmaxf(mx, t[e]), t[e + 1]);
                        mx1 = fmaxf(fmaxf(mx1, t[e + 2]), t[e + 3]);
                    }
                };
#pragma unroll
                for (int kp = 0; kp < 4; ++kp) {  // pairs (0,1) (2,3) (4,5) and the single tile 6
                    const int kt = 2 * kp;
                    if (kt + 2 < 7) read_k(kt + 2, kf[(kp + 1) & 1][0]);
                    if (kt + 3 < 7) read_k(kt + 3, kf[(kp + 1) & 1][1]);
                    f32x16 a, b;
#pragma unroll
                    for (int e = 0; e < 16; ++e) a[e] = b[e] = 0.f;
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) {
                        a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[kp & 1][0][ks], qf[ks], a, 0, 0, 0);
                        if (kt + 1 < 7) b = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[kp & 1][1][ks], qf[ks], b, 0, 0, 0);
                    }
                    s[kt] = a;
                    if (kt + 1 < 7) s[kt + 1] = b;
                    if (kp >= 1) {  // the previous pair is complete: its maximum rides under this pair's MFMAs
                        tile_max(s[kt - 2]);
                        tile_max(s[kt - 1]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                ATTN_STAMP(3)  // S^T (+ running maximum)
                mx = fmaxf(mx, mx1);
                // key of s[kt][e] = 32kt + (e&3) + 8(e>>2) + 4hh ; keys >= 197 are padding: of the last
                // tile only e = 0..3 can be valid (keys 192..195 in the lower lane half, 196 in the upper)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (e + 4 * hh < VIT_T - 192) mx = fmaxf(mx, s[6][e]);
                mx = fmaxf(mx, other_half(mx));
                const float nmx = -mx * sc;
                float sum4[4] = {0.f, 0.f, 0.f, 0.f};  // four independent chains: a single one serialises 104 dependent adds
                // P of step i (16 keys: 8 values per lane), exponentiated, summed and rounded to bf16
                auto soft = [&](auto i_tag, bf16x8& pf) {
                    constexpr int I = decltype(i_tag)::value;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        constexpr int kt = I >> 1;
                        const int e = 8 * (I & 1) + j;
                        float pv = (dbg & 8) ? s[kt][e] : __builtin_amdgcn_exp2f(fmaf(s[kt][e], sc, nmx));
                        if (kt == 6 && !(e < 4 && e + 4 * hh < VIT_T - 192)) pv = 0.f;
                        sum4[j & 3] += pv;
                        pf[j] = (bf16_t)pv;
                    }
                };
                ATTN_STAMP(4)  // maximum
#pragma unroll
                for (int e = 0; e < 16; ++e) o[0][e] = o[1][e] = 0.f;
                bf16x8 pf[2];
                read_v(0, vf[0]);
                soft(std::integral_constant<int, 0>{}, pf[0]);
                auto pv_step = [&](auto i_tag) {
                    constexpr int I = decltype(i_tag)::value;
                    if (I + 1 < 13) read_v(I + 1, vf[(I + 1) & 1]);
                    o[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, vf[I & 1][0]), pf[I & 1], o[0], 0, 0, 0);
                    if constexpr (I + 1 < 13) soft(std::integral_constant<int, (I + 1 < 13 ? I + 1 : 0)>{}, pf[(I + 1) & 1]);
                    o[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, vf[I & 1][1]), pf[I & 1], o[1], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                };
                pv_step(std::integral_constant<int, 0>{}); pv_step(std::integral_constant<int, 1>{});
                pv_step(std::integral_constant<int, 2>{}); pv_step(std::integral_constant<int, 3>{});
                pv_step(std::integral_constant<int, 4>{}); pv_step(std::integral_constant<int, 5>{});
                pv_step(std::integral_constant<int, 6>{}); pv_step(std::integral_constant<int, 7>{});
                pv_step(std::integral_constant<int, 8>{}); pv_step(std::integral_constant<int, 9>{});
                pv_step(std::integral_constant<int, 10>{}); pv_step(std::integral_constant<int, 11>{});
                pv_step(std::integral_constant<int, 12>{});
                float sum = (sum4[0] + sum4[1]) + (sum4[2] + sum4[3]);
                sum += other_half(sum);
                inv = __builtin_amdgcn_rcpf(sum);
            } else {
            // S^T tiles; the K fragments of tile kt+1 are requested before the MFMAs of tile kt so
            // that no MFMA waits on the ds_read issued just before it
            bf16x8 kf[2][4];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) kf[0][ks] = *(const bf16x8*)(Kl + r * ROWB + (((2 * ks + hh) ^ ksw) << 4));
#pragma unroll
            for (int kt = 0; kt < 7; ++kt) {
                if (kt + 1 < 7) {
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks)
                        kf[(kt + 1) & 1][ks] = *(const bf16x8*)(Kl + ((kt + 1) * 32 + r) * ROWB + (((2 * ks + hh) ^ ksw) << 4));
                }
                f32x16 a;
#pragma unroll
                for (int e = 0; e < 16; ++e) a[e] = 0.f;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[kt & 1][ks], qf[ks], a, 0, 0, 0);
                s[kt] = a;
                __builtin_amdgcn_sched_barrier(0);
            }
            ATTN_STAMP(3)  // S^T
            // key of s[kt][e] = 32kt + (e&3) + 8(e>>2) + 4hh ; keys >= 197 are padding: of the last
            // tile only e = 0..3 can be valid (keys 192..195 in the lower lane half, 196 in the upper)
            float mx = -INFINITY;
#pragma unroll
            for (int kt = 0; kt < 6; ++kt)
#pragma unroll
                for (int e = 0; e < 16; ++e) mx = fmaxf(mx, s[kt][e]);
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (e + 4 * hh < VIT_T - 192) mx = fmaxf(mx, s[6][e]);
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const f32x2 sc2 = {sc, sc}, nmx2 = {-mx * sc, -mx * sc};
            f32x2 sum2 = {0.f, 0.f};
#pragma unroll
            for (int kt = 0; kt < 6; ++kt)
#pragma unroll
                for (int e = 0; e < 16; e += 2) {
                    const f32x2 t = __builtin_elementwise_fma(f32x2{s[kt][e], s[kt][e + 1]}, sc2, nmx2);
                    const f32x2 p = {__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)};
                    s[kt][e] = p.x;
                    s[kt][e + 1] = p.y;
                    sum2 += p;
                }
#pragma unroll
            for (int e = 0; e < 4; e += 2) {
                const f32x2 t = __builtin_elementwise_fma(f32x2{s[6][e], s[6][e + 1]}, sc2, nmx2);
                f32x2 p = {__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)};
                if (!(e + 4 * hh < VIT_T - 192)) p.x = 0.f;
                if (!(e + 1 + 4 * hh < VIT_T - 192)) p.y = 0.f;
                s[6][e] = p.x;
                s[6][e + 1] = p.y;
                sum2 += p;
            }
#pragma unroll
            for (int e = 4; e < 16; ++e) s[6][e] = 0.f;
            float sum = sum2.x + sum2.y;
            sum += __shfl_xor(sum, 32, 64);
            inv = __builtin_amdgcn_rcpf(sum);

            ATTN_STAMP(4)  // softmax
#pragma unroll
            for (int e = 0; e < 16; ++e) o[0][e] = o[1][e] = 0.f;
            // 13 steps of 16 keys (keys 208..223 are all padding); the V^T fragments of step i+1 are
            // requested before the MFMAs of step i
            read_v(0, vf[0]);
#pragma unroll
            for (int i = 0; i < 13; ++i) {
                if (i + 1 < 13) read_v(i + 1, vf[(i + 1) & 1]);
                bf16x8 pf;
#pragma unroll
                for (int j = 0; j < 8; ++j) pf[j] = (bf16_t)s[i >> 1][8 * (i & 1) + j];
#pragma unroll
                for (int db = 0; db < 2; ++db)
                    o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, vf[i & 1][db]), pf, o[db], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            }
            ATTN_STAMP(5)  // P.V
            if (h + 1 < h_end) {  // before the stores: the wait for the prefetched Q must not cover them
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) qf[ks] = qn[ks];
            }
            // o[db][4*rg + j] = O[q][32db + 8rg + 4hh + j]: pair the lane halves into 16-byte stores
            bf16_t* op = out + ((size_t)b * VIT_T + min(q, VIT_T - 1)) * VIT_D + h * VIT_DH;
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int rp = 0; rp < 4; rp += 2) {
                    uint2 u0, u1;  // row groups rp and rp+1 of this lane
                    {
                        bf16x4 t0, t1;
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            t0[j] = (bf16_t)(o[db][rp * 4 + j] * inv);
                            t1[j] = (bf16_t)(o[db][(rp + 1) * 4 + j] * inv);
                        }
                        u0 = __builtin_bit_cast(uint2, t0);
                        u1 = __builtin_bit_cast(uint2, t1);
                    }
                    // lower half keeps group rp and receives the upper half's group rp (dims +4..+7);
                    // upper half keeps group rp+1 and receives the lower half's (dims +0..+3)
                    const auto ax = __builtin_amdgcn_permlane32_swap(u0.x, u1.x, false, false);
                    const auto ay = __builtin_amdgcn_permlane32_swap(u0.y, u1.y, false, false);
                    const uint4 w = make_uint4(ax[0], ay[0], ax[1], ay[1]);
                    if (q < VIT_T) *(uint4*)(op + db * 32 + (rp + hh) * 8) = w;
                }
            ATTN_STAMP(6)  // Q hand-over + stores issued
        }
    }
    if (STAMP && lane == 0 && stamps) {
        unsigned long long* o = stamps + ((size_t)blockIdx.x * 8 + wave) * 8;
#pragma unroll
        for (int i = 0; i < 7; ++i) o[i] = st[i];
        o[7] = (unsigned long long)(h_end - h_begin);
        if (wave == 7) {  // the staging wave has no compute phases: its slots 5 / 6 carry the clock pair of the workgroup
            o[5] = __builtin_amdgcn_s_memtime() - st_c0;
            o[6] = __builtin_amdgcn_s_memrealtime() - st_r0;
        }
    }
#undef ATTN_STAMP
}

}  // namespace

hipError_t launch_attention(const void* qkv, void* out, int B, hipStream_t s) {
    if (B <= 0) return hipSuccess;
    // MME_ATTN_BUFS=3: three K/V buffers (two heads in flight); measured 3 % slower than two, kept for A/B runs
    const char* nb_env = diag_env("MME_ATTN_BUFS");  // read per launch: an A/B run flips it inside one process
    const int nb = nb_env ? atoi(nb_env) : 2;
    // enough workgroups for two per CU-slot: split a crop's heads over 1, 2, 3, 4, 6 or 12 workgroups
    int hsplit = 1;
    for (int d : {1, 2, 3, 4, 6, 12}) {
        hsplit = d;
        if (B * d >= 512) break;
    }
    const char* pipe_env = diag_env("MME_ATTN_PIPE");
    if (pipe_env && atoi(pipe_env) == 0) {  // the phase-by-phase softmax (A/B)
        if (hipError_t e = ensure_dynamic_lds((const void*)attn_fwd_t197<2, false, false>, AttnGeom<2>::LDS_BYTES); e != hipSuccess) return e;
        hipLaunchKernelGGL((attn_fwd_t197<2, false, false>), dim3(B * hsplit), dim3(512), AttnGeom<2>::LDS_BYTES, s, (const bf16_t*)qkv, (bf16_t*)out, B, hsplit,
                           (unsigned long long*)nullptr);
        return hipGetLastError();
    }
    if (nb == 2) {
        if (hipError_t e = ensure_dynamic_lds((const void*)attn_fwd_t197<2>, AttnGeom<2>::LDS_BYTES); e != hipSuccess) return e;
        hipLaunchKernelGGL((attn_fwd_t197<2>), dim3(B * hsplit), dim3(512), AttnGeom<2>::LDS_BYTES, s, (const bf16_t*)qkv, (bf16_t*)out, B, hsplit,
                           (unsigned long long*)nullptr);
    } else {
        if (hipError_t e = ensure_dynamic_lds((const void*)attn_fwd_t197<3>, AttnGeom<3>::LDS_BYTES); e != hipSuccess) return e;
        hipLaunchKernelGGL((attn_fwd_t197<3>), dim3(B * hsplit), dim3(512), AttnGeom<3>::LDS_BYTES, s, (const bf16_t*)qkv, (bf16_t*)out, B, hsplit,
                           (unsigned long long*)nullptr);
    }
    return hipGetLastError();
}

// diagnostic: the stamped build (two buffers, one workgroup per crop); stamps = uint64[B][8 waves][8], zeroed by the caller
hipError_t launch_attention_stamped(const void* qkv, void* out, int B, unsigned long long* stamps, hipStream_t s) {
    const int dbg = diag_env("MME_ATTN_DEBUG") ? atoi(diag_env("MME_ATTN_DEBUG")) : 0;
    if (B <= 0) return hipSuccess;
    if (hipError_t e = ensure_dynamic_lds((const void*)attn_fwd_t197<2, true>, AttnGeom<2>::LDS_BYTES); e != hipSuccess) return e;
    hipLaunchKernelGGL((attn_fwd_t197<2, true>), dim3(B), dim3(512), AttnGeom<2>::LDS_BYTES, s, (const bf16_t*)qkv, (bf16_t*)out, B, 1, stamps, dbg);
    return hipGetLastError();
}
