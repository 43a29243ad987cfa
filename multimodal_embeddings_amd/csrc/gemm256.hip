// bf16 "TN" GEMM, 256 x 256 x 64 tile, 8 waves, 2-slot operand ring (variant 2; kept for A/B and for K < 128 --
// the default for large problems is gemm256r.hip, the same kernel with a 3-deep activation ring).
//
//   C[M,N] = A[M,K] . W[N,K]^T (+ fused epilogue), both operands K-contiguous.
//
// Design for CDNA4 (one workgroup of 512 threads per CU, 128 KiB of the 160 KiB LDS):
//   * 8 waves as 2 (M) x 4 (N); a wave owns 128 x 64 of C = 8 x 4 MFMA 16x16x32 tiles
//     (128 accumulator VGPRs).  Waves w and w+4 share a SIMD and belong to different M halves.
//   * The two M halves ("groups") run one barrier apart: while group 0 issues the 16 MFMAs of
//     a phase, group 1 reads its next fragments from LDS and issues LDS-DMA loads, and vice
//     versa, so each SIMD's matrix pipe always has one wave feeding it (ping-pong).
//   * A K-tile (64 deep) is 4 phases = the 4 quadrants (64 x 32) of the wave tile.  B fragments
//     of both column halves stay in registers for the whole K-tile, A fragments are re-read
//     per row half: 20 ds_read_b128 per wave per K-tile for 64 MFMAs.
//   * global -> LDS by LDS-DMA in HALF tiles (128 rows x 64 k = 16 KiB; 2 x 1 KiB pieces per
//     wave), into a 2-slot ring.  Each phase issues one half tile, ordered so that a region is
//     re-filled only after the last wave finished reading it (activations first: they come from
//     HBM / Infinity Cache and need the longest lead, the weights are L2 resident):
//         tile t, q0: A_lo(t+1)   q1: A_hi(t+1)   q2: B_hi(t+1)   q3: B_lo(t+2)
//     One counted wait per K-tile (`s_waitcnt vmcnt(2)` in q3: everything but the two newest
//     DMA pieces has landed), followed by a barrier a full phase before the first read of tile
//     t+1 -- LDS-DMA data is ordered for a ds_read only by the issuer's vmcnt + a barrier.
//     Loads stay in flight across barriers (raw s_barrier, never __syncthreads).
//   * LDS image: 128-byte rows, 16-byte chunk index XOR ((row>>1)&7): conflict-free
//     ds_read_b128 fragment reads; the swizzle is applied to the DMA SOURCE address (the LDS
//     destination of LDS-DMA is lane-linear).
//   * Persistent over tiles (grid = #CUs), XCD-aware tile order: the tiles that share an A row
//     panel run back to back on one XCD, so A is fetched from HBM once and served from L2.
//     The next tile's first K-tile is requested BEFORE the epilogue of the current one, so the
//     epilogue's loads/stores and the prologue latency overlap (K is only 12 K-tiles deep for
//     three of the four ViT GEMM shapes, so per-tile overhead matters as much as the main loop).
#include <cstdio>
#include <cstdlib>

#include "common.h"
#include "gemm_epilogue.h"
#include "kernels.h"

namespace {

constexpr int TM = 256, TN = 256, TK = 64;
constexpr int HALF = 128 * TK * 2;  // 16 KiB: 128 rows x 128 B
constexpr int SLOT = 4 * HALF;      // A_lo | A_hi | B_lo | B_hi
constexpr int R_ALO = 0, R_AHI = HALF, R_BLO = 2 * HALF, R_BHI = 3 * HALF;

#define S_BARRIER() asm volatile("s_barrier" ::: "memory")

template <int EPI>
__global__ __launch_bounds__(512, 2) void gemm_bf16_tn_256(GemmArgs g, int tiles_m, int tiles_n, int gn, int dbg) {
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
    const int b_base = R_BLO + (wn >> 1) * HALF + ((wn & 1) * 64 + fr) * 128;
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
        const int id = xcd_remap(tile, ntiles);
        const int gsz = tiles_m * gn;
        const int grp = id / gsz, rem = id - grp * gsz;
        const int gw = min(gn, tiles_n - grp * gn);
        const int tm = rem / gw, tn = grp * gn + (rem - tm * gw);
        c.m0 = tm * TM;
        c.n0 = tn * TN;
        c.Ag = (const char*)g.A + (size_t)(dbg == 2 ? 0 : c.m0) * ldb;
        c.Wg = (const char*)g.W + (size_t)(dbg == 2 ? 0 : c.n0) * ldb;
        c.mrem = g.M - 1 - c.m0;  // rows past the matrix edge re-read the last row
        c.nrem = g.N - 1 - c.n0;
        return c;
    };
    // one half tile = 128 rows: pieces 2*wave and 2*wave+1; `rem` clamps the row, `half` = 0/128
    auto stage = [&](const char* gbase, int rem, int half, int region) {
        glds16(gbase + (min(pr0 + half, rem) * ldb + pc0), dst0 + region);
        glds16(gbase + (min(pr1 + half, rem) * ldb + pc1), dst1 + region);
    };
    // K-tile 0 completely, plus B_lo of K-tile 1 (10 DMA pieces per wave)
    auto issue_prologue = [&](const TileCtx& c) {
        stage(c.Ag, c.mrem, 0, R_ALO);
        stage(c.Ag, c.mrem, 128, R_AHI);
        stage(c.Wg, c.nrem, 0, R_BLO);
        stage(c.Wg, c.nrem, 128, R_BHI);
        if (nk > 1) stage(c.Wg + TK * 2, c.nrem, 0, SLOT + R_BLO);
    };

    int tile = blockIdx.x;
    if (tile >= ntiles) return;
    TileCtx cx = make_ctx(tile);
    issue_prologue(cx);
    bool stores_pending = false;  // the previous tile's 16 epilogue stores may still be in flight

    while (true) {
        const int m0 = cx.m0, n0 = cx.n0, mrem = cx.mrem, nrem = cx.nrem;
        const char* Ag = cx.Ag;
        const char* Wg = cx.Wg;

        f32x4 acc[8][4];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

        // K-tile 0 must have landed; B_lo of K-tile 1 (2 pieces) and the previous tile's stores
        // (younger than every prologue piece: vmcnt retires in issue order) may stay in flight.
        if (stores_pending) {
            asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        } else if (nk > 1) {
            asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
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

#define K_TILE(cur, nxt)                                                                       \
    {                                                                                          \
        const bool has1 = t + 1 < nk && dbg != 1, has2 = t + 2 < nk && dbg != 1;               \
        const char* a1 = Ag + (size_t)(t + 1) * (TK * 2);                                      \
        const char* w1 = Wg + (size_t)(t + 1) * (TK * 2);                                      \
        /* q0 */                                                                               \
        READ_B(cur, 0)                                                                         \
        __builtin_amdgcn_sched_barrier(0);                                                     \
        READ_A(cur, 0)                                                                         \
        if (has1) stage(a1, mrem, 0, (nxt) + R_ALO);                                        \
        S_BARRIER();                                                                           \
        MFMA_QUAD(0, 0)                                                                        \
        S_BARRIER();                                                                           \
        /* q1 */                                                                               \
        READ_B(cur, 2)                                                                         \
        if (has1) stage(a1, mrem, 128, (nxt) + R_AHI);                                      \
        S_BARRIER();                                                                           \
        MFMA_QUAD(0, 2)                                                                        \
        S_BARRIER();                                                                           \
        /* q2 */                                                                               \
        READ_A(cur, 4)                                                                         \
        if (has1) stage(w1, nrem, 128, (nxt) + R_BHI);                                      \
        S_BARRIER();                                                                           \
        MFMA_QUAD(4, 2)                                                                        \
        S_BARRIER();                                                                           \
        /* q3 */                                                                               \
        if (has2) {                                                                            \
            stage(w1 + TK * 2, nrem, 0, (cur) + R_BLO);                                     \
            asm volatile("s_waitcnt vmcnt(2)" ::: "memory");                                   \
        } else {                                                                               \
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                   \
        }                                                                                      \
        S_BARRIER();                                                                           \
        MFMA_QUAD(4, 0)                                                                        \
        S_BARRIER();                                                                           \
    }

        int t = 0;
        for (; t + 1 < nk; t += 2) {
            K_TILE(0, SLOT)
            ++t;
            K_TILE(SLOT, 0)
            --t;
        }
        if (t < nk) K_TILE(0, SLOT)
        if (wm == 0) S_BARRIER();

        // Every LDS read of this tile is complete: stream the next tile's first K-tile in while
        // the epilogue below runs (its loads and stores are younger, see the wait above).
        const int next_tile = tile + gridDim.x;
        const bool has_next = next_tile < ntiles;
        if (has_next) {
            cx = make_ctx(next_tile);
            issue_prologue(cx);
        }
        bool interior = false;

        // epilogue: acc[i][j][r] is C[m0 + wm*128 + i*16 + fr][n0 + wn*64 + j*16 + fq*4 + r].
        // vmcnt counts stores too on CDNA4, so a load inside the store loop would wait for every
        // store issued before it: all loads (bias, residual, positions) are issued first, with
        // row indices clamped instead of branched, and the stores are fire-and-forget.
        if (EPI != EPI_F32 && n0 + TN <= g.N && m0 + TM <= g.M) {
            interior = true;
            // interior tile: straight-line code, no per-lane predicate (a branch would make the
            // compiler re-insert vmcnt(0) -- i.e. a wait for the stores -- at every join)
            if constexpr (EPI == EPI_PATCH) {
                const int nb = n0 + wn * 64 + fq * 4;
                const int mb = m0 + wm * 128 + fr;
                f32x4 bv[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) bv[j] = *(const f32x4*)(g.bias + nb + j * 16);
#pragma unroll
                for (int i0 = 0; i0 < 8; i0 += 2) {
                    EpiRow er[2];
                    f32x4 pv[2][4];
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        er[i] = epi_row<EPI>(mb + (i0 + i) * 16);
#pragma unroll
                        for (int j = 0; j < 4; ++j) pv[i][j] = *(const f32x4*)(g.pos + (int64_t)er[i].prow * g.N + nb + j * 16);
                    }
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const f32x4 v = acc[i0 + i][j] + bv[j] + pv[i][j];
                            *(uint2*)((bf16_t*)g.out + er[i].orow * g.ldo + nb + j * 16) = pack_bf16x4(v);
                        }
                }
            } else {
                if constexpr (epi_has_fast_path<EPI>()) epilogue_wave_128x64<EPI>(g, acc, m0 + wm * 128, n0 + wn * 64, fr, fq, [] {});
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
        if (!has_next) break;
        if (!interior) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // unknown store count: drain
        stores_pending = interior && EPI != EPI_PATCH;
        if (interior && EPI == EPI_PATCH) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        tile = next_tile;
    }
#undef READ_A
#undef READ_B
#undef MFMA_QUAD
#undef K_TILE
}

template <int EPI>
hipError_t launch256(const GemmArgs& g, hipStream_t s) {
    const int smem = 2 * SLOT;
    if (hipError_t e = ensure_dynamic_lds((const void*)gemm_bf16_tn_256<EPI>, smem); e != hipSuccess) return e;
    const int tiles_m = (g.M + TM - 1) / TM, tiles_n = (g.N + TN - 1) / TN;
    const int ntiles = tiles_m * tiles_n;
    const int grid = ntiles < 256 ? ntiles : 256;  // one workgroup per CU
    static const int dbg = [] {  // timing experiments only: 1 = no K-loop loads, 2 = every tile reads tile 0 (wrong results!)
        const int v = getenv("MME_GEMM_DEBUG") ? atoi(getenv("MME_GEMM_DEBUG")) : 0;
        if (v) fprintf(stderr, "libmme: MME_GEMM_DEBUG=%d -- GEMM RESULTS ARE INVALID (timing experiment mode)\n", v);
        return v;
    }();
    static const int gn_env = getenv("MME_GEMM_GN") ? atoi(getenv("MME_GEMM_GN")) : 0;
    // column-group width: the group's weight rows (gn x 256 x K bf16) should stay resident in one
    // XCD's 4 MiB L2 next to the streaming A panels and output lines; never split below 3 tiles
    // (PMC, fc1 4096 crops: L2-miss fetch 15.4 GB at gn = 12 -> 6.2 GB at gn = 6, same time)
    int gn = gn_env > 0 ? gn_env : (int)((2400 * 1024) / ((size_t)TN * g.K * 2));
    if (gn < 3) gn = 3;
    if (gn > tiles_n) gn = tiles_n;
    hipLaunchKernelGGL(gemm_bf16_tn_256<EPI>, dim3(grid), dim3(512), smem, s, g, tiles_m, tiles_n, gn, dbg);
    return hipGetLastError();
}

}  // namespace

hipError_t launch_gemm256(int epilogue, const GemmArgs& g, hipStream_t s) {
    if (g.M <= 0 || g.N <= 0) return hipSuccess;
    if (g.K <= 0 || (g.K % TK) != 0) return hipErrorInvalidValue;
    switch (epilogue) {
        case EPI_BIAS: return launch256<EPI_BIAS>(g, s);
        case EPI_BIAS_GELU: return launch256<EPI_BIAS_GELU>(g, s);
        case EPI_BIAS_RES: return launch256<EPI_BIAS_RES>(g, s);
        case EPI_PATCH: return launch256<EPI_PATCH>(g, s);
        case EPI_F32: return launch256<EPI_F32>(g, s);
        case EPI_LN_BIAS: return launch256<EPI_LN_BIAS>(g, s);
        case EPI_LN_BIAS_GELU: return launch256<EPI_LN_BIAS_GELU>(g, s);
        default: return hipErrorInvalidValue;
    }
}
