// bf16 "TN" GEMM, 256 x 256 x 64 tile, 8 waves, THREE-deep activation ring (variant 3, the default for large problems).
//
// 8 waves as 2 x 4 (wave tile 128 x 64), two wave groups one barrier apart (ping-pong), XOR-swizzled
// 128-byte LDS rows filled by LDS-DMA, epilogues of gemm_epilogue.h.  Staging schedule:  All 160 KiB of LDS are used: a 3-slot ring for the A (activation)
// half tiles and a 2-slot ring for the B (weight) half tiles.  A comes from HBM / Infinity
// Cache (an L2 miss costs well over a K-tile of time under load), B is L2 resident, so A is
// requested TWO K-tiles ahead and B one (across output tiles: the stream of K-tiles of a workgroup's
// consecutive tiles is one stream, see the tile loop):
//     tile t, q0: B_hi(t+1)   q1: A_lo(t+2)   q2: A_hi(t+2)   q3: B_lo(t+2), s_waitcnt vmcnt(6)
// vmcnt retires in issue order, so the counted wait in q3 covers B_hi(t+1) and everything older
// (A(t+1) was issued during tile t-1, B_lo(t+1) in its q3) and leaves the six newest pieces --
// A(t+2) and B_lo(t+2) -- in flight: every A piece has 1.5-1.75 K-tiles to land, B_hi three
// phases instead of one.
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "gemm_epilogue.h"
#include "kernels.h"

namespace {

constexpr int TM = 256, TN = 256, TK = 64;
constexpr int HALF = 128 * TK * 2;  // 16 KiB: 128 rows x 128 B
constexpr int ASLOT = 2 * HALF;     // A_lo | A_hi, ring of 3 at [0, 96 KiB)
constexpr int BSLOT = 2 * HALF;     // B_lo | B_hi, ring of 2 at [96 KiB, 160 KiB)
constexpr int B_RING = 3 * ASLOT;
constexpr int LDS_BYTES = 3 * ASLOT + 2 * BSLOT;

#define S_BARRIER() asm volatile("s_barrier" ::: "memory")

// STAMP: diagnostic build -- waves 0 and 4 accumulate s_memtime intervals between the eight barriers of
// a K-tile, the time inside the counted wait and the time between two tiles' K loops (split into group
// sync + prologue issue, epilogue body, rest); 16 words per wave go to `stamps` (layout: mme.h,
// mme_gemm_stamps).  No stamp executes in the product kernel.
//
// DEFER (variants 4 and 5): the output write of a 256 x 256 bf16 tile is a fixed ~8 k cycles of store ISSUE per CU
// (~16 B/clk/CU) during which the matrix pipe idles.  With DEFER an interior tile stores only the upper half
// of each wave tile at once; the packed lower half (8 x 16 B per lane, 32 VGPRs -- the K loop has that many
// to spare) is issued one store per phase inside the first K-tiles of the workgroup's NEXT tile, behind
// that phase's LDS-DMA request, where the other wave group's MFMAs cover it.  vmcnt counts stores, in order:
// the counted wait of such a K-tile leaves nine operations in flight instead of six.
// NDEF = 16-byte stores per lane that are deferred (0, 4, 6 or 8 of the 16), issued one or two per phase of the
// next tile's first K-tile.
template <int EPI, bool STAMP = false, int NDEF = 0>
__global__ __launch_bounds__(512, 2) void gemm_bf16_tn_256r(GemmArgs g, int tiles_m, int tiles_n, int gn, int dbg,
                                                             unsigned long long* stamps = nullptr, int rb = 0) {
    constexpr bool DEFER = NDEF > 0;
        static_assert(NDEF == 0 || NDEF == 4 || NDEF == 6 || NDEF == 8, "NDEF");
    static_assert(!DEFER || epi_has_fast_path<EPI>() || (EPI == EPI_F32 && NDEF == 4), "DEFER needs a 16-byte fast-path epilogue");
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int fr = lane & 15, fq = lane >> 4, fsw = (lane >> 1) & 7;
    const int ldb = g.K * 2;
    const int nk = g.K / TK;
    const int ntiles = tiles_m * tiles_n;

    // fragment read bases inside a slot
    const int a_base = wm * HALF + fr * 128;
    const int b_base = B_RING + (wn >> 1) * HALF + ((wn & 1) * 64 + fr) * 128;
    const int ch0 = ((0 + fq) ^ fsw) << 4, ch1 = ((4 + fq) ^ fsw) << 4;
    // DMA piece geometry of this lane: pieces 2*wave, 2*wave+1 of a half tile
    const int pr0 = (wave * 2) * 8 + (lane >> 3), pr1 = pr0 + 8;
    const int pc0 = ((lane & 7) ^ ((pr0 >> 1) & 7)) << 4, pc1 = ((lane & 7) ^ ((pr1 >> 1) & 7)) << 4;
    char* const dst0 = lds + (wave * 2) * 1024;
    char* const dst1 = dst0 + 1024;

    // Per-tile state is wave-uniform (SGPRs); the per-lane DMA source offset of a piece is
    // recomputed per use (one v_min + one v_mad) instead of living in 8 VGPRs across the epilogue.
    struct TileCtx {
        int m0, n0, mrem, nrem;
        const char *Ag, *Wg;
    };
    auto make_ctx = [&](int tile) {
        TileCtx c;
        // ids are ordered (column group, row panel, column in group): an XCD's contiguous id range
        // stays inside one group of `gn` column tiles, whose weight rows then live in its L2
        // rb > 0: the row panels are cut into blocks of rb and the column groups alternate INSIDE a block, so the second
        // group re-reads a block of A panels (rb x 256 rows) a few hundred tiles after the first read it -- from the
        // Infinity Cache -- instead of after the whole activation matrix has streamed through
        const int id = xcd_remap(tile, ntiles);
        const int rbs = rb > 0 ? rb : tiles_m;
        const int per_blk = rbs * tiles_n;
        const int blk = id / per_blk, idb = id - blk * per_blk;
        const int rows_blk = min(rbs, tiles_m - blk * rbs);
        const int gsz = rows_blk * gn;
        const int grp = idb / gsz, rem = idb - grp * gsz;
        const int gw = min(gn, tiles_n - grp * gn);
        const int tm_f = blk * rbs + rem / gw, tn = grp * gn + (rem - (rem / gw) * gw);
        const int tm = g.reverse_m ? tiles_m - 1 - tm_f : tm_f;
        c.m0 = tm * TM;
        c.n0 = tn * TN;
        c.Ag = (const char*)g.A + (size_t)((dbg & 3) == 2 ? 0 : c.m0) * ldb;
        c.Wg = (const char*)g.W + (size_t)((dbg & 3) == 2 ? 0 : c.n0) * ldb;
        c.mrem = g.M - 1 - c.m0;  // rows past the matrix edge re-read the last row
        c.nrem = g.N - 1 - c.n0;
        return c;
    };
    // one half tile = 128 rows: pieces 2*wave and 2*wave+1; `rem` clamps the row, `half` = 0/128
    // (buffer form of LDS-DMA: wave-uniform base in a resource descriptor + one 32-bit lane offset,
    // half the address traffic of the 64-bit global form)
    auto stage = [&](const char* gbase, int rem, int half, int region) {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)gbase, 0, 0x7fffffff, 0x00020000);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (LDS_AS void*)(dst0 + region), 16, min(pr0 + half, rem) * ldb + pc0, 0, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (LDS_AS void*)(dst1 + region), 16, min(pr1 + half, rem) * ldb + pc1, 0, 0, 0);
    };
    // K-tile 0 completely, then B_lo and A of K-tile 1 (14 DMA pieces per wave)
    auto issue_prologue = [&](const TileCtx& c) {
        stage(c.Ag, c.mrem, 0, 0);
        stage(c.Ag, c.mrem, 128, HALF);
        stage(c.Wg, c.nrem, 0, B_RING);
        stage(c.Wg, c.nrem, 128, B_RING + HALF);
        if (nk > 1) stage(c.Wg + TK * 2, c.nrem, 0, B_RING + BSLOT);
    };
    // ... and A of K-tile 1 (4 pieces); on interior tiles issued behind the epilogue's loads
    auto issue_prologue_a1 = [&](const TileCtx& c) {
        if (nk > 1) {
            stage(c.Ag + TK * 2, c.mrem, 0, ASLOT);
            stage(c.Ag + TK * 2, c.mrem, 128, ASLOT + HALF);
        }
    };

    unsigned long long st_iv[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_wait = 0, st_epi = 0, st_prev = 0, st_nk = 0, st_pro = 0, st_body = 0;
    const bool st_on = STAMP && (wave == 0 || wave == 4);
#define STAMP_IV(i)                                          \
    if (st_on) {                                             \
        const unsigned long long now = __builtin_amdgcn_s_memtime(); \
        st_iv[i] += now - st_prev;                           \
        st_prev = now;                                       \
    }
    int tile = blockIdx.x;
    if (tile >= ntiles) return;
    if (g.run_if && *g.run_if == 0) return;  // device-side switch of a fallback launch (uniform)
    // clock the chip holds under this kernel: shader cycles (s_memtime) per 100 MHz tick (s_memrealtime)
    const unsigned long long st_c0 = st_on ? __builtin_amdgcn_s_memtime() : 0, st_r0 = st_on ? __builtin_amdgcn_s_memrealtime() : 0;
    TileCtx cx = make_ctx(tile);
    issue_prologue(cx);
    issue_prologue_a1(cx);
    // The operand rings run on ACROSS output tiles: during the last two K-tiles of a tile the look-ahead
    // slots (B one K-tile ahead, A two) receive the first K-tiles of the workgroup's NEXT tile, so after
    // the first tile there is no prologue burst and no wait for a first K-tile: it landed, counted by the
    // ordinary q3 wait, before the epilogue started.
    // a_cur / a_nx2: LDS offsets of the A ring slots of K-tiles t and t+2; b_cur / b_nxt: B ring
    int a_cur = 0, a_nx2 = 2 * ASLOT, b_cur = 0, b_nxt = BSLOT;
    bool first_tile = true;
    // DEFER: rows 64..127 of the previous tile's wave tile, packed, and where they go (wave-uniform base)
    uint4 pend[DEFER ? NDEF : 1];
    bf16_t* pend_base = nullptr;
    const int pend_lo0 = fr * (int)g.ldo + row16_col(0, fq), pend_lo1 = fr * (int)g.ldo + row16_col(2, fq);
    auto pend_store = [&](auto idx_tag) {
        constexpr int IDX = decltype(idx_tag)::value;  // (i - (8 - NDEF / 2)) * 2 + (jp >> 1)
        if constexpr (DEFER && IDX < NDEF) {
            if constexpr (EPI == EPI_F32) {  // f32 out: the four column tiles j = IDX of the last row block (i = 7)
                float* fb = (float*)pend_base;
                *(uint4*)(fb + (int64_t)(7 * 16 + fr) * g.ldf + IDX * 16 + fq * 4) = pend[IDX];
            } else {
                *(uint4*)(pend_base + (int64_t)(8 - NDEF / 2 + (IDX >> 1)) * 16 * g.ldo + ((IDX & 1) ? pend_lo1 : pend_lo0)) = pend[IDX];
            }
        }
    };
    // pending stores issued in phase q of the carrying K-tile: 4 -> 1,1,1,1   6 -> 2,2,1,1   8 -> 2,2,2,2
    constexpr int PQ0 = NDEF >= 6 ? 2 : 1, PQ1 = NDEF >= 6 ? 2 : 1, PQ2 = NDEF >= 8 ? 2 : 1;
    constexpr int PQ_BEFORE_WAIT = PQ0 + PQ1 + PQ2;

    while (true) {
        const int m0 = cx.m0, n0 = cx.n0, mrem = cx.mrem, nrem = cx.nrem;
        const char* Ag = cx.Ag;
        const char* Wg = cx.Wg;
        const int next_tile = tile + gridDim.x;
        const bool has_next = next_tile < ntiles;
        const TileCtx nx = has_next ? make_ctx(next_tile) : cx;

        f32x4 acc[8][4];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

        // first tile only: K-tile 0 must have landed (B_lo and A of K-tile 1, 6 pieces, may stay in flight);
        // later tiles found their first K-tiles counted by the previous tile's last q3 wait
        if (first_tile) {
            asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            first_tile = false;
        }
        S_BARRIER();
        if (wm == 1) S_BARRIER();  // group 1 runs one barrier behind group 0

        bf16x8 fa[4][2], fb[4][2];

#define READ_B(slot_off, n_first)                                                              \
    _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                            \
        fb[(n_first) + j][0] = *(const bf16x8*)(lds + (slot_off) + b_base + ((n_first) + j) * 2048 + ch0); \
        fb[(n_first) + j][1] = *(const bf16x8*)(lds + (slot_off) + b_base + ((n_first) + j) * 2048 + ch1); \
    }
#define READ_A(slot_off, m_first)                                                              \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                            \
        fa[i][0] = *(const bf16x8*)(lds + (slot_off) + a_base + ((m_first) + i) * 2048 + ch0); \
        fa[i][1] = *(const bf16x8*)(lds + (slot_off) + a_base + ((m_first) + i) * 2048 + ch1); \
    }
#define MFMA_QUAD(m_first, n_first)                                                            \
    __builtin_amdgcn_s_setprio(1);                                                             \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                           \
    _Pragma("unroll") for (int i = 0; i < 4; ++i)                                              \
    _Pragma("unroll") for (int j = 0; j < 2; ++j)                                              \
        acc[(m_first) + i][(n_first) + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(          \
            fb[(n_first) + j][ks], fa[i][ks], acc[(m_first) + i][(n_first) + j], 0, 0, 0);     \
    __builtin_amdgcn_s_setprio(0);

        if (st_on) {
            const unsigned long long now = __builtin_amdgcn_s_memtime();
            if (st_prev) st_epi += now - st_prev;
            st_prev = now;
            st_nk += nk;
        }
        // one K-tile; PS >= 0: also issue pending stores 0..3 of the previous tile, one per phase (DEFER)
        auto ktile = [&](const int t, auto ps_tag) {
            constexpr int PS = decltype(ps_tag)::value;
            // K-tiles t+1 / t+2 of the stream: past the end of this tile they are the next tile's first ones
            const bool nt1 = t + 1 >= nk, nt2 = t + 2 >= nk;
            const bool has1 = (!nt1 || has_next) && (dbg & 3) != 1, has2 = (!nt2 || has_next) && (dbg & 3) != 1;
            const char* w1 = nt1 ? nx.Wg + (size_t)(t + 1 - nk) * (TK * 2) : Wg + (size_t)(t + 1) * (TK * 2);
            const char* w2 = nt2 ? nx.Wg + (size_t)(t + 2 - nk) * (TK * 2) : Wg + (size_t)(t + 2) * (TK * 2);
            const char* a2 = nt2 ? nx.Ag + (size_t)(t + 2 - nk) * (TK * 2) : Ag + (size_t)(t + 2) * (TK * 2);
            const int nrem1 = nt1 ? nx.nrem : nrem, nrem2 = nt2 ? nx.nrem : nrem, mrem2 = nt2 ? nx.mrem : mrem;
            /* q0 */
            READ_B(b_cur, 0)
            __builtin_amdgcn_sched_barrier(0);
            READ_A(a_cur, 0)
            if (has1) stage(w1, nrem1, 128, B_RING + b_nxt + HALF);
            if constexpr (PS >= 0) {
                pend_store(std::integral_constant<int, 0>{});
                if constexpr (PQ0 > 1) pend_store(std::integral_constant<int, 1>{});
            }
            S_BARRIER();
            STAMP_IV(0)
            MFMA_QUAD(0, 0)
            S_BARRIER();
            STAMP_IV(1)
            /* q1 */
            READ_B(b_cur, 2)
            if (has2) stage(a2, mrem2, 0, a_nx2);
            if constexpr (PS >= 0) {
                pend_store(std::integral_constant<int, PQ0>{});
                if constexpr (PQ1 > 1) pend_store(std::integral_constant<int, PQ0 + 1>{});
            }
            S_BARRIER();
            STAMP_IV(2)
            MFMA_QUAD(0, 2)
            S_BARRIER();
            STAMP_IV(3)
            /* q2 */
            READ_A(a_cur, 4)
            if (has2) stage(a2, mrem2, 128, a_nx2 + HALF);
            if constexpr (PS >= 0) {
                pend_store(std::integral_constant<int, PQ0 + PQ1>{});
                if constexpr (PQ2 > 1) pend_store(std::integral_constant<int, PQ0 + PQ1 + 1>{});
            }
            S_BARRIER();
            STAMP_IV(4)
            MFMA_QUAD(4, 2)
            S_BARRIER();
            STAMP_IV(5)
            /* q3 */
            const unsigned long long st_w0 = st_on ? __builtin_amdgcn_s_memtime() : 0;
            if (has2) {
                stage(w2, nrem2, 0, B_RING + b_cur);
                // leave in flight: A(t+2) and B_lo(t+2) (six pieces) -- and the deferred stores issued among them
                if constexpr (PS >= 0 && PQ_BEFORE_WAIT == 6) {
                    asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
                } else if constexpr (PS >= 0 && PQ_BEFORE_WAIT == 5) {
                    asm volatile("s_waitcnt vmcnt(11)" ::: "memory");
                } else if constexpr (PS >= 0) {
                    static_assert(PS < 0 || PQ_BEFORE_WAIT == 3, "counted wait");
                    asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
                } else {
                    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
                }
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            if (st_on) st_wait += __builtin_amdgcn_s_memtime() - st_w0;
            if constexpr (PS >= 0) {
                pend_store(std::integral_constant<int, PQ_BEFORE_WAIT>{});
                if constexpr (NDEF - PQ_BEFORE_WAIT > 1) pend_store(std::integral_constant<int, PQ_BEFORE_WAIT + 1>{});
            }
            S_BARRIER();
            STAMP_IV(6)
            MFMA_QUAD(4, 0)
            S_BARRIER();
            STAMP_IV(7)
            a_cur = a_cur == 2 * ASLOT ? 0 : a_cur + ASLOT;
            a_nx2 = a_nx2 == 2 * ASLOT ? 0 : a_nx2 + ASLOT;
            b_cur ^= BSLOT;
            b_nxt ^= BSLOT;
        };
        int t_first = 0;
        if (DEFER && pend_base != nullptr) {  // wave-uniform: the previous tile left its lower half pending
            // all deferred stores ride in the first K-tile (a second store-carrying instance of the body, or a
            // register rotation through one instance, spills 30+ VGPRs and loses more than it hides)
            ktile(0, std::integral_constant<int, 0>{});
            t_first = 1;
            pend_base = nullptr;
        }
        for (int t = t_first; t < nk; ++t) ktile(t, std::integral_constant<int, -1>{});
        if (wm == 0) S_BARRIER();

        unsigned long long st_t1 = 0;
        if (st_on) {
            st_t1 = __builtin_amdgcn_s_memtime();
            st_pro += st_t1 - st_prev;
        }

        // epilogue: acc[i][j][r] is C[m0 + wm*128 + i*16 + fr][n0 + wn*64 + j*16 + fq*4 + r].
        // vmcnt counts stores too on CDNA4, so a load inside the store loop would wait for every
        // store issued before it: all loads (bias, residual, positions) are issued first, with
        // row indices clamped instead of branched, and the stores are fire-and-forget.
        const bool fast = EPI != EPI_F32 && EPI != EPI_TOPK && n0 + TN <= g.N && m0 + TM <= g.M;
        // f32 out (K9 cosine block): interior tiles of a 16-byte-aligned, ld % 4 == 0 matrix store straight-line as well --
        // 32 fire-and-forget 16-byte stores per lane instead of 32 guarded ones with a wait after each
        const bool fast_f32 = EPI == EPI_F32 && n0 + TN <= g.N && m0 + TM <= g.M && (g.ldf & 3) == 0 && ((uintptr_t)g.outf & 15) == 0;
        if (EPI == EPI_F32 && fast_f32) {
            float* fb = g.outf + (int64_t)(m0 + wm * 128) * g.ldf + n0 + wn * 64;  // wave-uniform
            const int64_t lo = (int64_t)fr * g.ldf + fq * 4;
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (DEFER && i == 7) {
                        pend[j < (DEFER ? NDEF : 1) ? j : 0] = __builtin_bit_cast(uint4, acc[i][j]);
                    } else {
                        *(f32x4*)(fb + (int64_t)i * 16 * g.ldf + lo + j * 16) = acc[i][j];
                    }
                }
            if (DEFER) pend_base = (bf16_t*)fb;
        } else if (fast) {
            // interior tile: straight-line code, no per-lane predicate (a branch would make the
            // compiler re-insert vmcnt(0) -- i.e. a wait for the stores -- at every join)
            if constexpr (EPI == EPI_PATCH) {
                epilogue_wave_patch_128x64(g, acc, m0 + wm * 128, n0 + wn * 64, fr, fq, [] {});
            } else {
                if constexpr (DEFER && epi_has_fast_path<EPI>()) {
                    epilogue_wave_128x64<EPI, NDEF>(g, acc, m0 + wm * 128, n0 + wn * 64, fr, fq, [] {}, pend);
                    pend_base = (bf16_t*)g.out + (int64_t)(m0 + wm * 128) * g.ldo + (n0 + wn * 64);
                } else if constexpr (epi_has_fast_path<EPI>()) {
                    epilogue_wave_128x64<EPI>(g, acc, m0 + wm * 128, n0 + wn * 64, fr, fq, [] {});
                }
            }
        } else if constexpr (EPI == EPI_TOPK) {
            // candidate filter: the eight row thresholds of this lane are loaded once; a (row, 16-column)
            // fragment is looked at element by element only if one of its four values reaches the bar
            float tau[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int m = m0 + wm * 128 + i * 16 + fr;
                tau[i] = m < g.M ? g.thr[(int64_t)m * g.thr_stride] : INFINITY;
            }
            // slots are reserved with ONE returning atomic per (lane, row) -- all eight issued before any
            // result is needed -- instead of one round trip per hit
            const int nb = n0 + wn * 64 + fq * 4;
            int cnt[8], base[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                int h = 0;
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) h += (acc[i][j][r] >= tau[i] && nb + j * 16 + r < g.N) ? 1 : 0;
                cnt[i] = h;
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) base[i] = cnt[i] ? atomicAdd(g.cand_count + (m0 + wm * 128 + i * 16 + fr), cnt[i]) : 0;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (!cnt[i]) continue;
                const int64_t row = (int64_t)(m0 + wm * 128 + i * 16 + fr) * g.cand_cap;
                int k = base[i];
                if (k + cnt[i] > g.cand_cap) atomicOr(g.overflow, 1);
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (acc[i][j][r] >= tau[i] && nb + j * 16 + r < g.N) {
                            if (k < g.cand_cap) {
                                g.cand_val[row + k] = acc[i][j][r];
                                g.cand_idx[row + k] = nb + j * 16 + r;
                            }
                            ++k;
                        }
            }
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int m = m0 + wm * 128 + i * 16 + fr;
                if (m >= g.M) continue;
                const EpiRow er = epi_row<EPI>(m);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int n = n0 + wn * 64 + j * 16 + fq * 4;
                    if (n >= g.N) continue;
                    epi_store<EPI>(g, m, er, n, acc[i][j]);
                }
            }
        }
        if (st_on) st_body += __builtin_amdgcn_s_memtime() - st_t1;
        if (!has_next) {
            if constexpr (DEFER) {
                if (pend_base != nullptr) {
                    pend_store(std::integral_constant<int, 0>{}); pend_store(std::integral_constant<int, 1>{});
                    pend_store(std::integral_constant<int, 2>{}); pend_store(std::integral_constant<int, 3>{});
                    pend_store(std::integral_constant<int, 4>{}); pend_store(std::integral_constant<int, 5>{});
                    pend_store(std::integral_constant<int, 6>{}); pend_store(std::integral_constant<int, 7>{});
                }
            }
            break;
        }
        cx = nx;
        tile = next_tile;
    }
    if (st_on && lane == 0 && stamps) {
        unsigned long long* o = stamps + ((size_t)blockIdx.x * 2 + (wave >> 2)) * 16;
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = st_iv[i];
        o[8] = st_wait;
        o[9] = st_epi;
        o[10] = st_nk;
        o[11] = st_pro;
        o[12] = st_body;
        o[13] = __builtin_amdgcn_s_memtime() - st_c0;
        o[14] = __builtin_amdgcn_s_memrealtime() - st_r0;
    }
#undef STAMP_IV
#undef READ_A
#undef READ_B
#undef MFMA_QUAD
#undef K_TILE
}

template <int EPI, int DEFER = 0>
hipError_t launch256r(const GemmArgs& g, hipStream_t s) {
    const int smem = LDS_BYTES;
    if (hipError_t e = ensure_dynamic_lds((const void*)gemm_bf16_tn_256r<EPI, false, DEFER>, smem); e != hipSuccess) return e;
    const int tiles_m = (g.M + TM - 1) / TM, tiles_n = (g.N + TN - 1) / TN;
    const int ntiles = tiles_m * tiles_n;
    const int grid = ntiles < 256 ? ntiles : 256;  // one workgroup per CU
    static const int dbg = [] {  // timing experiments only: 1 = no K-loop loads, 2 = every tile reads tile 0 (wrong results!)
        const int v = diag_env("MME_GEMM_DEBUG") ? atoi(diag_env("MME_GEMM_DEBUG")) : 0;
        if (v) fprintf(stderr, "libmme: MME_GEMM_DEBUG=%d -- GEMM RESULTS ARE INVALID (timing experiment mode)\n", v);
        return v;
    }();
    static const int gn_env = diag_env("MME_GEMM_GN") ? atoi(diag_env("MME_GEMM_GN")) : 0;
    // column-group width: the group's weight rows (gn x 256 x K bf16) should stay resident in one
    // XCD's 4 MiB L2 next to the streaming A panels and output lines; never split below 3 tiles
    // (PMC, fc1 4096 crops: L2-miss fetch 15.4 GB at gn = 12 -> 6.2 GB at gn = 6, same time)
    int gn = gn_env > 0 ? gn_env : (int)((2400 * 1024) / ((size_t)TN * g.K * 2));
    if (gn < 3) gn = 3;
    if (gn > tiles_n) gn = tiles_n;
    // MME_GEMM_RB: row-panel block of the tile order (0 = one block: column group outermost, the default); read per launch for A/B runs
    const char* rb_env = diag_env("MME_GEMM_RB");
    // measured at 4096 crops: 16 -> -0.4 % GEMM time (8 / 32: +-0) but +5 % requests leaving the L2 (the weight group is
    // re-fetched per block; FETCH_SIZE counts Infinity-Cache hits too) -- inside the noise, so the default stays 0
    const int rb = rb_env ? atoi(rb_env) : 0;
    hipLaunchKernelGGL((gemm_bf16_tn_256r<EPI, false, DEFER>), dim3(grid), dim3(512), smem, s, g, tiles_m, tiles_n, gn, dbg,
                       (unsigned long long*)nullptr, rb);
    return hipGetLastError();
}

}  // namespace

// diagnostic launch of the stamped build (bias epilogue only): stamps = uint64[256 * 2 * 16], zeroed by the caller
hipError_t launch_gemm256r_stamped(const GemmArgs& g, unsigned long long* stamps, hipStream_t s) {
    if (g.M <= 0 || g.N <= 0 || g.K < 2 * TK || (g.K % TK) != 0) return hipErrorInvalidValue;
    hipError_t e = ensure_dynamic_lds((const void*)gemm_bf16_tn_256r<EPI_BIAS, true>, LDS_BYTES);
    if (e != hipSuccess) return e;
    const int tiles_m = (g.M + TM - 1) / TM, tiles_n = (g.N + TN - 1) / TN;
    const int ntiles = tiles_m * tiles_n;
    int gn = (int)((2400 * 1024) / ((size_t)TN * g.K * 2));
    gn = gn < 3 ? 3 : gn;
    gn = gn > tiles_n ? tiles_n : gn;
    // MME_GEMM_GRID: run the stamped build on fewer workgroups (does the epilogue's store cost depend on how
    // many CUs store at the same time?)
    int grid = ntiles < 256 ? ntiles : 256;
    if (diag_env("MME_GEMM_GRID") && atoi(diag_env("MME_GEMM_GRID")) > 0 && atoi(diag_env("MME_GEMM_GRID")) < grid) grid = atoi(diag_env("MME_GEMM_GRID"));
    hipLaunchKernelGGL((gemm_bf16_tn_256r<EPI_BIAS, true>), dim3(grid), dim3(512), LDS_BYTES, s, g, tiles_m, tiles_n, gn, 0, stamps);
    return hipGetLastError();
}

hipError_t launch_gemm256r(int epilogue, const GemmArgs& g, hipStream_t s, int defer) {
    if (g.M <= 0 || g.N <= 0) return hipSuccess;
    if (g.K <= 0 || (g.K % TK) != 0) return hipErrorInvalidValue;
    if (g.K < 2 * TK) return hipErrorInvalidValue;  // the cross-tile stream looks two K-tiles ahead (launch_gemm routes K = 64 to the 128 x 128 kernel)
    if (defer == 8) {
        switch (epilogue) {
            case EPI_BIAS: return launch256r<EPI_BIAS, 8>(g, s);
            case EPI_BIAS_GELU: return launch256r<EPI_BIAS_GELU, 8>(g, s);
            case EPI_BIAS_RES: return launch256r<EPI_BIAS_RES, 8>(g, s);
            case EPI_BIAS_RES_STATS: return launch256r<EPI_BIAS_RES_STATS, 8>(g, s);
            case EPI_LN_BIAS: return launch256r<EPI_LN_BIAS, 8>(g, s);
            case EPI_LN_BIAS_GELU: return launch256r<EPI_LN_BIAS_GELU, 8>(g, s);
            default: break;  // the other epilogues have no deferred form
        }
    } else if (defer == 6) {
        switch (epilogue) {
            case EPI_BIAS: return launch256r<EPI_BIAS, 6>(g, s);
            case EPI_BIAS_GELU: return launch256r<EPI_BIAS_GELU, 6>(g, s);
            case EPI_BIAS_RES: return launch256r<EPI_BIAS_RES, 6>(g, s);
            case EPI_BIAS_RES_STATS: return launch256r<EPI_BIAS_RES_STATS, 6>(g, s);
            case EPI_LN_BIAS: return launch256r<EPI_LN_BIAS, 6>(g, s);
            case EPI_LN_BIAS_GELU: return launch256r<EPI_LN_BIAS_GELU, 6>(g, s);
            default: break;
        }
    } else if (defer == 4) {
        if (epilogue == EPI_F32) return launch256r<EPI_F32, 4>(g, s);
        switch (epilogue) {
            case EPI_BIAS: return launch256r<EPI_BIAS, 4>(g, s);
            case EPI_BIAS_GELU: return launch256r<EPI_BIAS_GELU, 4>(g, s);
            case EPI_BIAS_RES: return launch256r<EPI_BIAS_RES, 4>(g, s);
            case EPI_BIAS_RES_STATS: return launch256r<EPI_BIAS_RES_STATS, 4>(g, s);
            case EPI_LN_BIAS: return launch256r<EPI_LN_BIAS, 4>(g, s);
            case EPI_LN_BIAS_GELU: return launch256r<EPI_LN_BIAS_GELU, 4>(g, s);
            default: break;
        }
    }
    switch (epilogue) {
        case EPI_BIAS: return launch256r<EPI_BIAS>(g, s);
        case EPI_BIAS_GELU: return launch256r<EPI_BIAS_GELU>(g, s);
        case EPI_BIAS_RES: return launch256r<EPI_BIAS_RES>(g, s);
        case EPI_BIAS_RES_STATS: return launch256r<EPI_BIAS_RES_STATS>(g, s);
        case EPI_PATCH: return launch256r<EPI_PATCH>(g, s);
        case EPI_F32: return launch256r<EPI_F32>(g, s);
        case EPI_LN_BIAS: return launch256r<EPI_LN_BIAS>(g, s);
        case EPI_LN_BIAS_GELU: return launch256r<EPI_LN_BIAS_GELU>(g, s);
        case EPI_TOPK: return launch256r<EPI_TOPK>(g, s);
        default: return hipErrorInvalidValue;
    }
}
