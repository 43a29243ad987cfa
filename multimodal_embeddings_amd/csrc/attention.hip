// K5: fused multi-head self-attention for ViT-B/16 (T = 197 tokens, 12 heads, dh = 64).
//
// Restates transformers models/vit/modeling_vit.py:164-189 (softmax(Q K^T / 8) V, softmax in
// f32).  T is short, so there is no online softmax: the whole 32 x 224 score strip of a query
// block lives in accumulator registers.
//
// One 8-wave workgroup per CU, PERSISTENT: it walks its blocks (a block = the 12 heads of one crop; 12 / hsplit heads
// for small batches) as one stream of (crop, head) items:
//   * K and V of item i+1 stream into the second LDS buffer by LDS-DMA (global_load_lds, no VGPR round trip) while
//     item i is computed, and the Q fragments of item i+1 are prefetched into registers -- across crop boundaries too:
//     after the workgroup's first item no memory latency is exposed.  One barrier per item.
//   * waves 0..6 own the 7 query blocks of 32 (197 -> 224); wave 7 has none.  The 50 LDS-DMA pieces of an item are
//     requested by wave 7 (18), wave 3 -- its SIMD partner -- (8) and waves 4..6 (8 each; ATTN_SHARE_*): one issuer alone needs
//     ~8 000 cycles per item for them (an LDS-DMA piece stalls its issuer ~160 cycles), which was the kernel's
//     critical path in round 2 (round 3: 14.2 -> 11.6 ms per step from spreading the requests alone).
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
constexpr int ATTN_SHARE_FAST = 0 + 16 * 8 + 256 * 8;  // K/V pieces of the next head requested by the computing waves (dma_head: a + 16 b + 256 c)
constexpr int ATTN_SHARE_EXACT = 0 + 16 * 8 + 256 * 8;
// NB = LDS buffers (heads in flight + 1): two buffers of 224 rows (7 key tiles of 32).  (A three-buffer geometry of 208
// rows, two heads in flight, measured 3 % slower in round 2 and is gone.)
template <int NB> struct AttnGeom {
    static_assert(NB == 2, "two K/V buffers");
    static constexpr int TROWS = 224;
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
// PIPE: 1 = exact row maximum, the softmax riding inside the two MFMA loops; 2 = FAST, the reference point of the
// exponentials taken from key tile 0 (see there), guarded; the exact form re-runs a launch whose guard was raised.
// run_if: when given, the whole launch returns at once unless *run_if != 0 (the conditional exact re-run).
// share: K/V pieces of the next head the COMPUTING waves request (encoding: dma_head); the staging wave takes the rest
template <int NB, bool STAMP, int PIPE>
__device__ __forceinline__ void attn_body(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out, const int nblk, const int hsplit,
                                          unsigned long long* stamps, const int dbg_arg, int* __restrict__ guard, const int share,
                                          const float guard_limit, const int only_block) {
    // ablation switches of the STAMPED build only (results invalid): 1 no K re-reads, 2 no K/V requests after the
    // first head, 4 no maximum, 8 no exponentials
    const int dbg = STAMP ? dbg_arg : 0;
    constexpr int TROWS = AttnGeom<NB>::TROWS, KV_BYTES = AttnGeom<NB>::KV_BYTES, BUF_BYTES = AttnGeom<NB>::BUF_BYTES;
    extern __shared__ __attribute__((aligned(16))) char lds[];  // NB x (K | V)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // A block = hpw = 12 / hsplit heads of one crop (hsplit blocks share a crop: small batches would otherwise leave
    // most CUs idle and serialise 12 heads on one).  The workgroup is PERSISTENT: it walks the blocks blockIdx.x,
    // blockIdx.x + gridDim.x, ... as ONE stream of (crop, head) items, so that the K/V requests and the Q prefetch of
    // a crop's first head ride under the previous crop's last head -- one exposed memory latency per workgroup
    // instead of one per crop (16 per CU at 4096 crops: ~3 % of the launch).
    const int hpw = VIT_H / hsplit;
    const int nloc = blockIdx.x < nblk ? (nblk - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;
    const int n_items = nloc * hpw;
    if (n_items == 0) return;
    // item -> (crop base, head)
    // rev (share bit 17): walk the blocks from the LAST crop down (zig-zag order of consecutive kernels, capi.hip forward_chunk)
    const bool rev = (share >> 17) & 1;
    auto item_blk = [&](int it) { const int blk = blockIdx.x + (it / hpw) * gridDim.x; return rev ? nblk - 1 - blk : blk; };
    auto item_head = [&](int it) { return (item_blk(it) % hsplit) * hpw + (it - (it / hpw) * hpw); };
    auto item_crop = [&](int it) { return item_blk(it) / hsplit; };

    // V rows 200.. are never written by DMA: zero them once in every buffer (P is 0 there, but
    // 0 * garbage could be NaN).  Rows 197..199 receive clamped copies of row 196 (finite).
    constexpr int ZROWS = TROWS - 200;
    for (int i = tid; i < NB * ZROWS * 8; i += 512) {
        const int buf = i / (ZROWS * 8), r = (i >> 3) % ZROWS, c = i & 7;
        *(uint4*)(lds + buf * BUF_BYTES + KV_BYTES + (200 + r) * ROWB + c * 16) = make_uint4(0, 0, 0, 0);
    }

    // DMA of one head: 25 K pieces + 25 V pieces (8 rows x 128 B each).  An LDS-DMA piece stalls its issuer for
    // ~100-160 cycles, so ONE issuer needs ~8 000 cycles per head for the 50 pieces -- more than a head's arithmetic
    // in the FAST form.  The wave without a query block (7) takes most of them; each computing wave requests `share`
    // pieces right after the barrier (pieces w, w + 7, ...), where its SIMD partner's work covers the stall.
    auto dma_head = [&](int it, int buf) {
        const char* hb = (const char*)qkv + (size_t)item_crop(it) * VIT_T * QKV_LD + item_head(it) * ROWB;
        char* kdst = lds + buf * BUF_BYTES;
        if ((dbg & 2) && it != 0) return;
        // share = a + 16 b + 256 c: waves 0-2 (the first-dispatched half: they lose the issue arbitration to their SIMD
        // partners 4-6, which run at priority 1) request a pieces each, waves 4-6 b each, wave 3 (whose SIMD partner is
        // the staging wave) c, the staging wave the rest
        const int sa = share & 15, sb = (share >> 4) & 15, sc3 = (share >> 8) & 63;
        int p_begin, p_end, p_step = 3;
        if (wave < 3) { p_begin = wave; p_end = 3 * sa; }
        else if (wave == 3) { p_begin = 3 * (sa + sb); p_end = p_begin + sc3; p_step = 1; }
        else if (wave < 7) { p_begin = 3 * sa + wave - 4; p_end = 3 * (sa + sb); }
        else { p_begin = 3 * (sa + sb) + sc3; p_end = 2 * NPIECE; p_step = 1; }
        for (int p = p_begin; p < p_end; p += p_step) {
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
    // PRESCALED: the Q columns of the QKV projection already carry sc (folded into W_q / b_q when the weights are
    // uploaded, capi.hip), so a score leaves the matrix pipe in log2 units
    constexpr bool PRESCALED = true;
    const int q = wave * 32 + r;                      // this lane's query (waves 0..6)
    // only_block >= 0: only that query block is computed and stored (the last layer when nothing but the pooled token's row
    // is read afterwards, mme_set_forward_pruning); the other waves still take their share of the K/V requests
    const bool active = wave < 7 && (only_block < 0 || wave == only_block);
    // this lane's Q row of an item: 4 x 16 bytes at + ks * 32
    auto q_ptr = [&](int it) {
        return (const char*)qkv + ((size_t)item_crop(it) * VIT_T + min(q, VIT_T - 1)) * QKV_LD + hh * 16 + item_head(it) * ROWB;
    };

    bf16x8 qf[4], qn[4];
    dma_head(0, 0);
    if (active) {
        const char* qp = q_ptr(0);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) qf[ks] = *(const bf16x8*)(qp + ks * 32);
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
    for (int it = 0; it < n_items; ++it) {
        const int buf = it & 1;
        const int h = item_head(it), b = item_crop(it);
        const char* Kl = lds + buf * BUF_BYTES;
        const char* Vl = Kl + KV_BYTES;
        // head h has landed (each wave waits for its own pieces), everybody is done with head h-1
        // (the 4 output stores of head h-1 are this wave's youngest vector-memory operations and may
        // stay in flight: vmcnt retires in order and counts stores)
        if (!active) {
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // the staging wave: this item's pieces have landed
        } else if (it == 0) {
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
        }
        ATTN_STAMP(0)  // own requests landed
        S_BARRIER();
        ATTN_STAMP(1)  // everybody's
        if (it + 1 < n_items) {
            dma_head(it + 1, buf ^ 1);
            if (active) {
                const char* qp = q_ptr(it + 1);
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) qn[ks] = *(const bf16x8*)(qp + ks * 32);
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
            if constexpr (PIPE == 2) {
                // FAST form.  softmax(s) = exp2(s' - c) / sum for ANY reference point c (s' = score in log2 units): the exact
                // row maximum is one choice, and the only thing it buys is range -- precision of exp2 does not depend on the
                // magnitude of its result, and bf16 P has f32's exponent range.  Here c is the maximum over the FIRST key tile
                // only (keys 0..31, the class token among them).  It is known before the other six tiles are multiplied, so
                // those start from accumulators holding -c: K.Q^T - c leaves the matrix pipe ready for exp2 -- no subtraction,
                // no scale (Q arrives pre-multiplied by dh^-0.5 log2 e: folded into W_q at load time), no 112-entry
                // maximum chain on the critical path.  5 vector instructions per score become 3.5.
                // Range: the sum is >= 1 (the tile-0 maximum contributes exp2(0)); a row whose true maximum exceeds c by
                // more than ~100 (raw scores 550 apart) would overflow -- such a row raises `guard` and the launch is redone
                // by the exact kernel (launch_attention), so every finite input gets the exact algorithm's result.
                bf16x8 kf[2][2][4];
                auto read_k = [&](int kt, bf16x8 (&dst)[4]) {
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) dst[ks] = *(const bf16x8*)(Kl + (kt * 32 + r) * ROWB + (((2 * ks + hh) ^ ksw) << 4));
                };
                read_k(0, kf[0][0]);
                read_k(1, kf[1][0]);
                read_k(2, kf[1][1]);
                {
                    f32x16 a;
#pragma unroll
                    for (int e = 0; e < 16; ++e) a[e] = 0.f;
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[0][0][ks], qf[ks], a, 0, 0, 0);
                    s[0] = a;
                }
                float mx = fmaxf(fmaxf(s[0][0], s[0][1]), s[0][2]), mx1 = fmaxf(fmaxf(s[0][3], s[0][4]), s[0][5]);
#pragma unroll
                for (int e = 6; e < 14; e += 4) {  // elements 6..13; 14 and 15 below (every index < 16: ADVICE r3)
                    mx = fmaxf(fmaxf(mx, s[0][e]), s[0][e + 1]);
                    mx1 = fmaxf(fmaxf(mx1, s[0][e + 2]), s[0][e + 3]);
                }
                mx = fmaxf(mx, s[0][14]);
                mx1 = fmaxf(mx1, s[0][15]);
                mx = fmaxf(mx, mx1);
                mx = fmaxf(mx, other_half(mx));
                f32x16 negm;
#pragma unroll
                for (int e = 0; e < 16; ++e) negm[e] = -mx;
#pragma unroll
                for (int kp = 0; kp < 3; ++kp) {  // pairs (1,2) (3,4) (5,6)
                    const int kt = 2 * kp + 1;
                    if (kt + 2 < 7) read_k(kt + 2, kf[kp & 1][0]);
                    if (kt + 3 < 7) read_k(kt + 3, kf[kp & 1][1]);
                    f32x16 a = negm, b = negm;
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) {
                        a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[(kp + 1) & 1][0][ks], qf[ks], a, 0, 0, 0);
                        b = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[(kp + 1) & 1][1][ks], qf[ks], b, 0, 0, 0);
                    }
                    s[kt] = a;
                    s[kt + 1] = b;
                    __builtin_amdgcn_sched_barrier(0);
                }
                ATTN_STAMP(3)  // S^T
                float sum4[4] = {0.f, 0.f, 0.f, 0.f};
                auto soft = [&](auto i_tag, bf16x8& pf) {
                    constexpr int I = decltype(i_tag)::value;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        constexpr int kt = I >> 1;
                        const int e = 8 * (I & 1) + j;
                        float pv = __builtin_amdgcn_exp2f(kt == 0 ? s[kt][e] - mx : s[kt][e]);
                        if (kt == 6 && !(e < 4 && e + 4 * hh < VIT_T - 192)) pv = 0.f;
                        sum4[j & 3] += pv;  // hipcc packs neighbouring chains into v_pk_add_f32: measured better here than 104 single adds (11.29 vs 11.54 ms per step)
                        pf[j] = (bf16_t)pv;
                    }
                };
                ATTN_STAMP(4)
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
                // !(sum < 2^100) also catches inf and NaN; one lane's word per offending row is enough
                if (guard && !(sum < guard_limit) && q < VIT_T) *guard = 1;  // guard_limit = 2^100 (0.5 in the forced-re-run test mode)
                inv = __builtin_amdgcn_rcpf(sum);
            } else if constexpr (PIPE == 1) {
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
                const float nmx = PRESCALED ? -mx : -mx * sc;
                float sum4[4] = {0.f, 0.f, 0.f, 0.f};  // four independent chains: a single one serialises 104 dependent adds
                // P of step i (16 keys: 8 values per lane), exponentiated, summed and rounded to bf16
                auto soft = [&](auto i_tag, bf16x8& pf) {
                    constexpr int I = decltype(i_tag)::value;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        constexpr int kt = I >> 1;
                        const int e = 8 * (I & 1) + j;
                        float pv = (dbg & 8) ? s[kt][e] : __builtin_amdgcn_exp2f(PRESCALED ? s[kt][e] + nmx : fmaf(s[kt][e], sc, nmx));
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
            }
            ATTN_STAMP(5)  // P.V
            if (it + 1 < n_items) {  // before the stores: the wait for the prefetched Q must not cover them
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
        o[7] = (unsigned long long)n_items;
        if (wave == 7) {  // the staging wave has no compute phases: its slots 5 / 6 carry the clock pair of the workgroup
            o[5] = __builtin_amdgcn_s_memtime() - st_c0;
            o[6] = __builtin_amdgcn_s_memrealtime() - st_r0;
        }
    }
#undef ATTN_STAMP
}

// run_if: when given, the whole launch returns at once unless *run_if != 0 (the conditional exact re-run behind FAST)
template <int NB, bool STAMP = false, int PIPE = 1>
__global__ __launch_bounds__(512, 2) void attn_fwd_t197(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out, int nblk, int hsplit,
                                                         unsigned long long* stamps = nullptr, int dbg_arg = 0, int* __restrict__ guard = nullptr,
                                                         int share = 0, const int* __restrict__ run_if = nullptr, float guard_limit = 1.2676506e30f,
                                                         int only_block = -1) {
    if (run_if && *(const volatile int*)run_if == 0) return;  // uniform: every wave of every workgroup takes the same way
    attn_body<NB, STAMP, PIPE>(qkv, out, nblk, hsplit, stamps, dbg_arg, guard, share, guard_limit, only_block);
}

}  // namespace

hipError_t launch_attention(const void* qkv, void* out, int B, hipStream_t s, int* guard, bool force_redo, int only_block, bool reverse) {
    if (only_block < -1 || only_block > 6) return hipErrorInvalidValue;
    if (B <= 0) return hipSuccess;
    // blocks of 12 / hsplit heads: enough of them for every CU (one workgroup fits per CU: 112 KiB of LDS)
    int hsplit = 1;
    for (int d : {1, 2, 3, 4, 6, 12}) {
        hsplit = d;
        if (B * d >= 512) break;
    }
    const int nblk = B * hsplit, grid = nblk < 256 ? nblk : 256;  // persistent: one workgroup per CU walks its blocks
    const char* pipe_env = diag_env("MME_ATTN_PIPE");  // 1: the exact kernel only (A/B)
    const bool fast = guard != nullptr && !(pipe_env && atoi(pipe_env) == 1);
    const char* share_env = diag_env("MME_ATTN_SHARE");  // K/V pieces requested by the computing waves (dma_head)
    int share = share_env ? atoi(share_env) : (fast ? ATTN_SHARE_FAST : ATTN_SHARE_EXACT);
    if (share < 0 || 3 * ((share & 15) + ((share >> 4) & 15)) + (share >> 8) > 2 * NPIECE) return hipErrorInvalidValue;
    if (reverse) share |= 1 << 17;
    if (hipError_t e = ensure_dynamic_lds((const void*)attn_fwd_t197<2>, AttnGeom<2>::LDS_BYTES); e != hipSuccess) return e;
    if (fast) {
        // FAST kernel, then the exact kernel on the same launch geometry, which returns at once unless a row of the fast
        // kernel left the range its reference point covers (*guard raised; guard is zeroed by the caller per pass)
        if (hipError_t e = ensure_dynamic_lds((const void*)attn_fwd_t197<2, false, 2>, AttnGeom<2>::LDS_BYTES); e != hipSuccess) return e;
        hipLaunchKernelGGL((attn_fwd_t197<2, false, 2>), dim3(grid), dim3(512), AttnGeom<2>::LDS_BYTES, s, (const bf16_t*)qkv, (bf16_t*)out, nblk, hsplit,
                           (unsigned long long*)nullptr, 0, guard, share, (const int*)nullptr, force_redo ? 0.5f : 1.2676506e30f, only_block);
        if (hipError_t e = hipGetLastError(); e != hipSuccess) return e;
        hipLaunchKernelGGL((attn_fwd_t197<2>), dim3(grid), dim3(512), AttnGeom<2>::LDS_BYTES, s, (const bf16_t*)qkv, (bf16_t*)out, nblk, hsplit,
                           (unsigned long long*)nullptr, 0, (int*)nullptr, ATTN_SHARE_EXACT | (reverse ? 1 << 17 : 0), (const int*)guard, 1.2676506e30f, only_block);
        return hipGetLastError();
    }
    hipLaunchKernelGGL((attn_fwd_t197<2>), dim3(grid), dim3(512), AttnGeom<2>::LDS_BYTES, s, (const bf16_t*)qkv, (bf16_t*)out, nblk, hsplit,
                       (unsigned long long*)nullptr, 0, (int*)nullptr, share, (const int*)nullptr, 1.2676506e30f, only_block);
    return hipGetLastError();
}

// diagnostic: the stamped build (two buffers, one workgroup per crop); stamps = uint64[B][8 waves][8], zeroed by the caller
hipError_t launch_attention_stamped(const void* qkv, void* out, int B, unsigned long long* stamps, hipStream_t s) {
    const int dbg = diag_env("MME_ATTN_DEBUG") ? atoi(diag_env("MME_ATTN_DEBUG")) : 0;
    if (B <= 0) return hipSuccess;
    const char* pipe_env = diag_env("MME_ATTN_PIPE");
    const bool fast = !(pipe_env && atoi(pipe_env) == 1);
    const char* share_env = diag_env("MME_ATTN_SHARE");
    const int share = share_env ? atoi(share_env) : (fast ? ATTN_SHARE_FAST : ATTN_SHARE_EXACT);
    if (share < 0 || 3 * ((share & 15) + ((share >> 4) & 15)) + (share >> 8) > 2 * NPIECE) return hipErrorInvalidValue;
    if (!fast) {
        if (hipError_t e = ensure_dynamic_lds((const void*)attn_fwd_t197<2, true>, AttnGeom<2>::LDS_BYTES); e != hipSuccess) return e;
        hipLaunchKernelGGL((attn_fwd_t197<2, true>), dim3(B), dim3(512), AttnGeom<2>::LDS_BYTES, s, (const bf16_t*)qkv, (bf16_t*)out, B, 1, stamps, dbg, (int*)nullptr, share, (const int*)nullptr);
    } else {
        if (hipError_t e = ensure_dynamic_lds((const void*)attn_fwd_t197<2, true, 2>, AttnGeom<2>::LDS_BYTES); e != hipSuccess) return e;
        hipLaunchKernelGGL((attn_fwd_t197<2, true, 2>), dim3(B), dim3(512), AttnGeom<2>::LDS_BYTES, s, (const bf16_t*)qkv, (bf16_t*)out, B, 1, stamps, dbg, (int*)nullptr, share, (const int*)nullptr);
    }
    return hipGetLastError();
}
