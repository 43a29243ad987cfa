// bf16 "TN" GEMM, 256 x 128 x 32 tile, 4 waves, TWO workgroups per CU.
//
//   C[M,N] = A[M,K] . W[N,K]^T (+ fused epilogue), both operands K-contiguous.
//
// Why this shape.  Three of the four ViT GEMMs are only K = 768 deep, so a 256x256 output tile
// spends 12 K-tiles in the MFMA loop and then an epilogue (store-issue bound, plus erf-GELU or
// a residual read) during which, in an 8-wave workgroup that owns the whole CU, every matrix
// pipe idles.  Here a workgroup is 4 waves (one per SIMD, 2 x 2 of 128 x 64 = the same
// per-wave tile and accumulators as gemm256*.hip) and uses 72 KiB of LDS, so TWO workgroups
// share a CU and run out of phase: while one stores / applies GELU, the other one's MFMAs keep
// all four SIMDs busy, and the VALU work of an epilogue runs beside the other's MFMAs.
//
// Schedule (per wave; software-pipelined K-tile stream as in gemm256s.hip):
//   * K-tiles are 32 deep (one MFMA k-step): A 16 KiB + W 8 KiB per K-tile, 3-slot LDS ring;
//     a K-tile is requested by LDS-DMA 2 K-tiles (8 phases) before its barrier;
//   * 4 phases of 8 MFMAs per K-tile = (row half, column half) of the wave tile; fragments are
//     read at least one phase ahead into ping-pong registers (56 VGPRs).  Two K-tiles (even, odd):
//        ph  MFMA(m,n) A    B      LDS reads issued                              DMA pieces
//        1   (0,0)     aP   bN0a   aQ <- A(m1)(t)                                1 of W(t+2)
//        2   (0,1)     aP   bN1                                                  1 of W(t+2)
//        -- lgkmcnt(0), vmcnt(6+) ; s_barrier : K-tile t+1 visible, slot of t free --
//        3   (1,1)     aQ   bN1    aP <- A(m0)(t+1)                              2 of A(t+3)
//        4   (1,0)     aQ   bN0a   bN0b <- B(n0)(t+1)                            2 of A(t+3)
//        5   (0,0)     aP   bN0b   bN1 <- B(n1)(t+1), aQ <- A(m1)(t+1)           1 of W(t+3)
//        6   (0,1)     aP   bN1                                                  1 of W(t+3)
//        -- barrier : K-tile t+2 visible, slot of t+1 free --
//        7   (1,1)     aQ   bN1    aP <- A(m0)(t+2), bN0a <- B(n0)(t+2)          2 of A(t+4)
//        8   (1,0)     aQ   bN0b   bN1 <- B(n1)(t+2)                             2 of A(t+4)
//     (DMA of K-tile t+3 goes to the slot K-tile t just left.)
//   * the K-tile stream continues across output tiles; at a tile boundary the epilogue issues
//     its loads, then the DMA of that step, then its 16 16-byte stores (vmcnt retires in order).
//   * LDS image: 64-byte rows; the 16-byte chunk index is XORed with 3 when bit 3 of the row is
//     set, which makes the 16-row x 64-byte fragment reads (ds_read_b128) conflict-free; the
//     swizzle is applied to the DMA source address.
#include "common.h"
#include "gemm_epilogue.h"
#include "kernels.h"

namespace {

constexpr int TM = 256, TN = 128, TK = 32;
constexpr int A_BYTES = TM * TK * 2;  // 16 KiB
constexpr int B_BYTES = TN * TK * 2;  // 8 KiB
constexpr int SLOT = A_BYTES + B_BYTES;
constexpr int NSLOT = 3;

#define S_BARRIER() asm volatile("s_barrier" ::: "memory")
#define PHASE_FENCE() asm volatile("" ::: "memory")

template <int EPI>
__global__ __launch_bounds__(256, 2) void gemm_bf16_tn_256x128(GemmArgs g, int tiles_m, int tiles_n) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int fr = lane & 15, fq = lane >> 4;
    const int ldb = g.K * 2;
    const int nk = g.K / TK;  // even (launcher)
    const int ntiles = tiles_m * tiles_n;
    if ((int)blockIdx.x >= ntiles) return;
    const int my_tiles = (ntiles - 1 - (int)blockIdx.x) / (int)gridDim.x + 1;
    const int S = my_tiles * nk;

    // fragment read: row r of a tile region at r*64, chunk c at ((c ^ (r&8 ? 3 : 0)) << 4)
    const int fchunk = (fq ^ ((fr & 8) ? 3 : 0)) << 4;
    const int a_rd = (wm * 128 + fr) * 64 + fchunk;            // + mt*1024 + slot
    const int b_rd = A_BYTES + (wn * 64 + fr) * 64 + fchunk;   // + nt*1024 + slot

    struct Cursor {
        int tile, kt, m0, n0, mrem, nrem;
        const char *Ag, *Wg;
        bool valid;
    };
    auto seek = [&](Cursor& c, int tile) {
        c.tile = tile;
        c.kt = 0;
        c.valid = tile < ntiles;
        const int id = xcd_remap(c.valid ? tile : 0, ntiles);
        const int tm = id / tiles_n, tn = id - tm * tiles_n;
        c.m0 = tm * TM;
        c.n0 = tn * TN;
        c.Ag = (const char*)g.A + (size_t)c.m0 * ldb;
        c.Wg = (const char*)g.W + (size_t)c.n0 * ldb;
        c.mrem = g.M - 1 - c.m0;
        c.nrem = g.N - 1 - c.n0;
    };
    auto advance = [&](Cursor& c) {
        if (++c.kt == nk) seek(c, c.tile + (int)gridDim.x);
    };
    // One DMA piece = 16 rows x 64 B.  Piece p of a region covers rows 16p..16p+15; lane -> row
    // 16p + (lane>>2), 16-byte slot lane&3 holding logical chunk slot ^ (row&8 ? 3 : 0).
    // Geometry is recomputed from the lane id per use (see gemm256s.hip).
    auto piece = [&](const char* gbase, int rem, int p, char* dst) {
        int ln = lane;
        asm volatile("" : "+v"(ln));
        const int r = p * 16 + (ln >> 2);
        const int c = ((ln & 3) ^ ((r & 8) ? 3 : 0)) << 4;
        glds16(gbase + (min(r, rem) * ldb + c), dst + p * 1024);
    };
    // A: 16 pieces (4 per wave: pieces 4*wave .. 4*wave+3), W: 8 pieces (2 per wave)
    auto dma_A2 = [&](const Cursor& c, int slot_off, int half) {  // 2 of this wave's 4 A pieces
        if (!c.valid) return;
        const char* p = c.Ag + (size_t)c.kt * (TK * 2);
        piece(p, c.mrem, wave * 4 + half * 2, lds + slot_off);
        piece(p, c.mrem, wave * 4 + half * 2 + 1, lds + slot_off);
    };
    auto dma_W1 = [&](const Cursor& c, int slot_off, int which) {  // 1 of this wave's 2 W pieces
        if (!c.valid) return;
        piece(c.Wg + (size_t)c.kt * (TK * 2), c.nrem, wave * 2 + which, lds + slot_off + A_BYTES);
    };
    auto dma_all = [&](const Cursor& c, int slot_off) {
        dma_A2(c, slot_off, 0);
        dma_A2(c, slot_off, 1);
        dma_W1(c, slot_off, 0);
        dma_W1(c, slot_off, 1);
    };

    bf16x8 aP[4], aQ[4], bN0a[2], bN0b[2], bN1[2];
    f32x4 acc[8][4];

#define READ_A(dstf, slot_off, m_first)                                                            \
    _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                  \
        dstf[i] = *(const bf16x8*)(lds + (slot_off) + a_rd + ((m_first) + i) * 1024);
#define READ_B(dstf, slot_off, n_first)                                                            \
    _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                  \
        dstf[j] = *(const bf16x8*)(lds + (slot_off) + b_rd + ((n_first) + j) * 1024);
#define MFMA8(fA, fB, m_first, n_first)                                                            \
    _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                  \
    _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                  \
        acc[(m_first) + i][(n_first) + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(              \
            fB[j], fA[i], acc[(m_first) + i][(n_first) + j], 0, 0, 0);

    // stream cursors: cc = K-tile being computed (for the epilogue), ld = next K-tile to request
    Cursor cc, ld;
    seek(cc, blockIdx.x);
    ld = cc;
    // LDS slot offsets of K-tiles t, t+1, t+2 (rotating); the slot of t is re-filled with t+3
    int s0 = 0, s1 = SLOT, s2 = 2 * SLOT;
    // prologue: K-tiles 0, 1, 2 (6 pieces each per wave)
    dma_all(ld, s0);
    advance(ld);
    if (S > 1) dma_all(ld, s1); else ld.valid = false;
    advance(ld);
    if (S > 2) dma_all(ld, s2); else ld.valid = false;
    advance(ld);
    if (S > 2) {
        asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    } else if (S > 1) {
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    S_BARRIER();
    READ_A(aP, s0, 0)
    READ_B(bN0a, s0, 0)
    READ_B(bN1, s0, 2)
    bool pending_stores = false;  // an interior epilogue's 16 stores may still be in flight
    bool dma_ahead = false;       // the DMA of the first phases after an epilogue was issued inside it

    // vmcnt at a barrier: the 6 pieces of the newest requested K-tile (and, right after an
    // epilogue, its 16 stores) may stay in flight
#define SYNC_POINT(NEWEST_REQUESTED)                                                               \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                             \
    if (NEWEST_REQUESTED) {                                                                        \
        if (pending_stores) asm volatile("s_waitcnt vmcnt(22)" ::: "memory");                      \
        else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");                                      \
    } else { /* end of the stream: no younger DMA to leave in flight */                            \
        if (pending_stores) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");                      \
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                      \
    }                                                                                              \
    pending_stores = false;                                                                        \
    S_BARRIER();

    auto epilogue = [&](int fill_slot) __attribute__((always_inline)) {
        const int m0 = cc.m0, n0 = cc.n0;
        if (epi_has_fast_path<EPI>() && n0 + TN <= g.N && m0 + TM <= g.M) {
            if constexpr (epi_has_fast_path<EPI>()) epilogue_wave_128x64<EPI>(g, acc, m0 + wm * 128, n0 + wn * 64, fr, fq, [&] { dma_all(ld, fill_slot); });
            pending_stores = true;
        } else {
            dma_all(ld, fill_slot);
            PHASE_FENCE();
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int m = m0 + wm * 128 + i * 16 + fr;
                const EpiRow er = epi_row<EPI>(min(m, g.M - 1));
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int n = n0 + wn * 64 + j * 16 + fq * 4;
                    if (m < g.M && n < g.N) epi_store<EPI>(g, m, er, n, acc[i][j]);
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // unknown store count: drain
            pending_stores = false;
        }
    };

    // Two K-tiles (t even in s0, t+1 in s1, t+2 in s2).  LAST: t+1 is the last K-tile of an
    // output tile: no pre-read across the epilogue, and the DMA of phases 7/8/1'/2' is issued
    // inside the epilogue instead.
#define K_PAIR(LAST)                                                                               \
    {                                                                                              \
        /* 1 */                                                                                    \
        READ_A(aQ, s0, 4)                                                                          \
        if (!dma_ahead) dma_W1(ld, s2, 0);                                                         \
        MFMA8(aP, bN0a, 0, 0)                                                                      \
        PHASE_FENCE();                                                                             \
        /* 2 */                                                                                    \
        if (!dma_ahead) { dma_W1(ld, s2, 1); advance(ld); }                                        \
        dma_ahead = false;                                                                         \
        MFMA8(aP, bN1, 0, 2)                                                                       \
        SYNC_POINT(st + 2 < S)                                                                     \
        /* 3: K-tile t+1 visible; slot s0 free -> K-tile t+3 */                                    \
        READ_A(aP, s1, 0)                                                                          \
        dma_A2(ld, s0, 0);                                                                         \
        MFMA8(aQ, bN1, 4, 2)                                                                       \
        PHASE_FENCE();                                                                             \
        /* 4 */                                                                                    \
        READ_B(bN0b, s1, 0)                                                                        \
        dma_A2(ld, s0, 1);                                                                         \
        MFMA8(aQ, bN0a, 4, 0)                                                                      \
        PHASE_FENCE();                                                                             \
        /* 5 */                                                                                    \
        READ_B(bN1, s1, 2)                                                                         \
        READ_A(aQ, s1, 4)                                                                          \
        dma_W1(ld, s0, 0);                                                                         \
        MFMA8(aP, bN0b, 0, 0)                                                                      \
        PHASE_FENCE();                                                                             \
        /* 6 */                                                                                    \
        dma_W1(ld, s0, 1);                                                                         \
        advance(ld);                                                                               \
        MFMA8(aP, bN1, 0, 2)                                                                       \
        SYNC_POINT(st + 3 < S)                                                                     \
        /* 7: K-tile t+2 visible; slot s1 free -> K-tile t+4 */                                    \
        if (!(LAST)) {                                                                             \
            READ_A(aP, s2, 0)                                                                      \
            READ_B(bN0a, s2, 0)                                                                    \
            dma_A2(ld, s1, 0);                                                                     \
        }                                                                                          \
        MFMA8(aQ, bN1, 4, 2)                                                                       \
        PHASE_FENCE();                                                                             \
        /* 8 */                                                                                    \
        if (!(LAST)) {                                                                             \
            READ_B(bN1, s2, 2)                                                                     \
            dma_A2(ld, s1, 1);                                                                     \
        }                                                                                          \
        MFMA8(aQ, bN0b, 4, 0)                                                                      \
        PHASE_FENCE();                                                                             \
        /* rotate: (t, t+1, t+2) -> (t+2, t+3, t+4) = slots (s2, s0, s1) */                        \
        st += 2;                                                                                   \
        {                                                                                          \
            const int o0 = s0, o1 = s1;                                                            \
            s0 = s2;                                                                               \
            s1 = o0;                                                                               \
            s2 = o1;                                                                               \
        }                                                                                          \
    }

    int st = 0;  // stream index of the K-tile in slot s0
    dma_ahead = true; // the prologue requested K-tiles 0..2 completely: nothing to finish in the first 1/2
    for (int ti = 0; ti < my_tiles; ++ti) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int kt = 0; kt + 2 < nk; kt += 2) K_PAIR(false)
        K_PAIR(true)
        // after the rotation: s0 = slot of the next tile's K-tile 0, s1 = K-tile 1 (both requested),
        // s2 = slot just freed by the last K-tile -> the whole K-tile "2" of the next tile goes there
        epilogue(s2);
        advance(ld);
        dma_ahead = true;
        seek(cc, cc.tile + (int)gridDim.x);
        if (ti + 1 < my_tiles) {
            READ_A(aP, s0, 0)
            READ_B(bN0a, s0, 0)
            READ_B(bN1, s0, 2)
        }
    }
#undef READ_A
#undef READ_B
#undef MFMA8
#undef K_PAIR
#undef SYNC_POINT
}

template <int EPI>
hipError_t launch256x128(const GemmArgs& g, hipStream_t s) {
    static bool attr_set = false;
    const int smem = NSLOT * SLOT;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm_bf16_tn_256x128<EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    const int tiles_m = (g.M + TM - 1) / TM, tiles_n = (g.N + TN - 1) / TN;
    const int ntiles = tiles_m * tiles_n;
    const int grid = ntiles < 512 ? ntiles : 512;  // two workgroups per CU
    hipLaunchKernelGGL(gemm_bf16_tn_256x128<EPI>, dim3(grid), dim3(256), smem, s, g, tiles_m, tiles_n);
    return hipGetLastError();
}

}  // namespace

hipError_t launch_gemm256x128(int epilogue, const GemmArgs& g, hipStream_t s) {
    if (g.M <= 0 || g.N <= 0) return hipSuccess;
    if (g.K <= 0 || (g.K % (2 * TK)) != 0) return hipErrorInvalidValue;
    switch (epilogue) {
        case EPI_BIAS: return launch256x128<EPI_BIAS>(g, s);
        case EPI_BIAS_GELU: return launch256x128<EPI_BIAS_GELU>(g, s);
        case EPI_BIAS_RES: return launch256x128<EPI_BIAS_RES>(g, s);
        case EPI_PATCH: return launch256x128<EPI_PATCH>(g, s);
        case EPI_F32: return launch256x128<EPI_F32>(g, s);
        case EPI_LN_BIAS: return launch256x128<EPI_LN_BIAS>(g, s);
        case EPI_LN_BIAS_GELU: return launch256x128<EPI_LN_BIAS_GELU>(g, s);
        default: return hipErrorInvalidValue;
    }
}
