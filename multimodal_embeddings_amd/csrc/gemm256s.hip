// bf16 "TN" GEMM, 256 x 256 x 64 tile, 8 waves, software-pipelined K-tile STREAM.
//
//   C[M,N] = A[M,K] . W[N,K]^T (+ fused epilogue), both operands K-contiguous.
//
// Same tile geometry, LDS image and LDS-DMA staging as gemm256.hip (see there), but a
// different schedule.  The ViT shapes are short in K (12 K-tiles for three of the four GEMMs),
// so what matters is that nothing ever drains:
//   * the K-tiles of ALL output tiles a workgroup owns form one stream s = 0,1,2,...; LDS-DMA
//     runs two K-tiles ahead of the MFMAs straight across output-tile boundaries, so the next
//     tile's operands arrive while the epilogue of the current tile stores;
//   * inside a wave, fragments are read from LDS at least one sub-phase BEFORE the MFMAs that
//     consume them, into ping-pong registers, so MFMAs never wait on LDS and all 8 waves issue
//     MFMAs all the time (the two waves of a SIMD share its matrix pipe);
//   * ONE barrier per K-tile.  A K-tile is 8 sub-phases of 8 MFMAs = (row half, column half,
//     k half) of the wave's 128 x 64 x 64 work, ordered so every fragment quarter is read once:
//        k    MFMA (m,n,ks)   A regs  B regs   LDS reads issued            DMA issued
//        1    (0,0,0)         aP      bN0a     aQ <- A(m1,ks0)
//        2    (0,1,0)         aP      bN1
//        3    (1,1,0)         aQ      bN1      aP <- A(m0,ks1)
//        4    (1,0,0)         aQ      bN0a     bN0b <- B(n0,ks1)
//        5    (0,0,1)         aP      bN0b     bN1 <- B(n1,ks1), aQ <- A(m1,ks1)
//        6    (0,1,1)         aP      bN1
//        -- s_waitcnt lgkmcnt(0), vmcnt(*) ; s_barrier: K-tile s+1 visible, slot of s free --
//        7    (1,1,1)         aQ      bN1      aP <- A(m0,ks0)(s+1), bN0a <- B(n0,ks0)(s+1)   A(s+2)
//        8    (1,0,1)         aQ      bN0b     bN1 <- B(n1,ks0)(s+1)                          B(s+2)
//     (the whole K-tile s+2 is requested right behind the barrier that frees its slot: what
//     bounds this kernel is the L2 -> LDS feed, ~64 KiB per K-tile per CU, so bytes in flight count)
//     56 fragment VGPRs + 128 accumulator VGPRs.
//   * every wave waits for its own DMA pieces (vmcnt) before the barrier that publishes a
//     K-tile, and that barrier precedes the first read of it; DMA of K-tile s+2 targets the slot
//     of K-tile s only after the barrier that follows the last read of s (sub-phase 5).
//   * at an output-tile boundary the epilogue's loads are issued BEFORE the DMA of that step and
//     its stores after it: vmcnt retires in order, so the epilogue never waits for DMA it does
//     not need, and the next barrier's wait (vmcnt(16)) leaves exactly the 16 stores in flight.
#include "common.h"
#include "gemm_epilogue.h"
#include "kernels.h"

namespace {

constexpr int TM = 256, TN = 256, TK = 64;
constexpr int HALF = 128 * TK * 2;
// LDS map (one base register per operand reaches both slots with 16-bit immediates):
//   [A_lo s0][A_hi s0][A_lo s1][A_hi s1] [B_lo s0][B_hi s0][B_lo s1][B_hi s1]
constexpr int SLOT = 2 * HALF;  // slot stride inside each operand's 64 KiB
constexpr int R_ALO = 0, R_AHI = HALF, R_BLO = 4 * HALF, R_BHI = 5 * HALF;

#define S_BARRIER() asm volatile("s_barrier" ::: "memory")
#define PHASE_FENCE() asm volatile("" ::: "memory")

template <int EPI>
__global__ __launch_bounds__(512, 2) void gemm_bf16_tn_256s(GemmArgs g, int tiles_m, int tiles_n, unsigned long long* stamps) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int fr = lane & 15, fq = lane >> 4, fsw = (lane >> 1) & 7;
    const int ldb = g.K * 2;
    const int nk = g.K / TK;
    const int ntiles = tiles_m * tiles_n;
    if ((int)blockIdx.x >= ntiles) return;
    const int my_tiles = (ntiles - 1 - (int)blockIdx.x) / (int)gridDim.x + 1;
    const int S = my_tiles * nk;  // K-tiles in this workgroup's stream

    const int a_base = wm * HALF + fr * 128;
    const int b_base = R_BLO + (wn >> 1) * HALF + ((wn & 1) * 64 + fr) * 128;
    const int ch0 = ((0 + fq) ^ fsw) << 4, ch1 = ((4 + fq) ^ fsw) << 4;
    char* const dst0 = lds + (wave * 2) * 1024;
    char* const dst1 = dst0 + 1024;

    struct Cursor {  // wave-uniform position in the K-tile stream
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
    // The per-lane piece geometry is recomputed from the lane id at every use (a handful of
    // VALU ops): kept in registers across the K loop it gets spilled, and a scratch reload is a
    // vector-memory operation whose wait would drain the DMA pipeline.
    auto stage = [&](const char* gbase, int rem, int half, int region) {
        int ln = lane;
        asm volatile("" : "+v"(ln));
        const int r0 = (wave * 2) * 8 + (ln >> 3), r1 = r0 + 8;
        const int c0 = ((ln & 7) ^ ((r0 >> 1) & 7)) << 4, c1 = ((ln & 7) ^ ((r1 >> 1) & 7)) << 4;
        glds16(gbase + (min(r0 + half, rem) * ldb + c0), dst0 + region);
        glds16(gbase + (min(r1 + half, rem) * ldb + c1), dst1 + region);
    };
    auto load_A = [&](const Cursor& c, int slot_off) {
        if (!c.valid) return;
        const char* p = c.Ag + (size_t)c.kt * (TK * 2);
        stage(p, c.mrem, 0, slot_off + R_ALO);
        stage(p, c.mrem, 128, slot_off + R_AHI);
    };
    auto load_Blo = [&](const Cursor& c, int slot_off) {
        if (c.valid) stage(c.Wg + (size_t)c.kt * (TK * 2), c.nrem, 0, slot_off + R_BLO);
    };
    auto load_Bhi = [&](const Cursor& c, int slot_off) {
        if (c.valid) stage(c.Wg + (size_t)c.kt * (TK * 2), c.nrem, 128, slot_off + R_BHI);
    };

    bf16x8 aP[4], aQ[4], bN0a[2], bN0b[2], bN1[2];
    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

#define READ_A(dstf, slot_off, m_first, chx)                                                         \
    _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                    \
        dstf[i] = *(const bf16x8*)(lds + (slot_off) + a_base + ((m_first) + i) * 2048 + (chx));
#define READ_B(dstf, slot_off, n_first, chx)                                                         \
    _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                    \
        dstf[j] = *(const bf16x8*)(lds + (slot_off) + b_base + ((n_first) + j) * 2048 + (chx));
#define MFMA8(fA, fB, m_first, n_first)                                                              \
    _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                    \
    _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                    \
        acc[(m_first) + i][(n_first) + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(                \
            fB[j], fA[i], acc[(m_first) + i][(n_first) + j], 0, 0, 0);

    Cursor cc, l1, l2;
    seek(cc, blockIdx.x);
    l1 = cc;
    // prologue: K-tile 0 -> slot 0, K-tile 1 -> slot 1 (16 DMA pieces per wave)
    load_A(l1, 0);
    load_Blo(l1, 0);
    load_Bhi(l1, 0);
    advance(l1);
    l2 = l1;
    if (S > 1) {
        load_A(l1, SLOT);
        load_Blo(l1, SLOT);
        load_Bhi(l1, SLOT);
        advance(l2);
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    } else {
        l1.valid = false;
        l2.valid = false;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    S_BARRIER();
    READ_A(aP, 0, 0, ch0)
    READ_B(bN0a, 0, 0, ch0)
    READ_B(bN1, 0, 2, ch0)
    bool pending_stores = false;  // an interior epilogue's 32 stores may still be in flight

    constexpr int PEND = 16;
    // Interior-tile epilogue: [bias/residual loads] [DMA of K-tile s+2] [16 stores].  The stores
    // are then exactly the PEND youngest vector-memory operations of the wave.
    auto epilogue = [&](int nxt_slot_unused, int cur_slot) {
        const int m0 = cc.m0, n0 = cc.n0;
        if (epi_has_fast_path<EPI>() && n0 + TN <= g.N && m0 + TM <= g.M) {
            if constexpr (epi_has_fast_path<EPI>()) epilogue_wave_128x64<EPI>(g, acc, m0 + wm * 128, n0 + wn * 64, fr, fq, [&] {
                load_A(l2, cur_slot);
                load_Blo(l2, cur_slot);
                load_Bhi(l2, cur_slot);
            });
            pending_stores = true;
        } else {
            load_A(l2, cur_slot);
            load_Blo(l2, cur_slot);
            load_Bhi(l2, cur_slot);
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
            pending_stores = false;  // unknown store count: the next barrier's vmcnt(0) drains them
        }
    };

    // One K-tile; CUR / NXT = LDS slot offsets of K-tiles s and s+1.  LAST = last K-tile of an
    // output tile: no pre-read of the next tile's fragments (the epilogue needs the registers)
    // and no DMA in sub-phases 7/8 (the epilogue issues it after its own loads).
#define K_TILE(CUR, NXT, LAST)                                                                       \
    {                                                                                                \
        /* 1 */                                                                                      \
        READ_A(aQ, CUR, 4, ch0)                                                                      \
        MFMA8(aP, bN0a, 0, 0)                                                                        \
        PHASE_FENCE();                                                                               \
        /* 2 */                                                                                      \
        MFMA8(aP, bN1, 0, 2)                                                                         \
        PHASE_FENCE();                                                                               \
        /* 3 */                                                                                      \
        READ_A(aP, CUR, 0, ch1)                                                                      \
        MFMA8(aQ, bN1, 4, 2)                                                                         \
        PHASE_FENCE();                                                                               \
        /* 4 */                                                                                      \
        READ_B(bN0b, CUR, 0, ch1)                                                                    \
        MFMA8(aQ, bN0a, 4, 0)                                                                        \
        PHASE_FENCE();                                                                               \
        /* 5 */                                                                                      \
        READ_B(bN1, CUR, 2, ch1)                                                                     \
        READ_A(aQ, CUR, 4, ch1)                                                                      \
        MFMA8(aP, bN0b, 0, 0)                                                                        \
        PHASE_FENCE();                                                                               \
        /* 6 */                                                                                      \
        MFMA8(aP, bN1, 0, 2)                                                                         \
        STAMP(t0)                                                                                    \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                           \
        STAMP(t1)                                                                                    \
        if (pending_stores) {                                                                        \
            static_assert(PEND == 16, "vmcnt literal below");                                        \
            asm volatile("s_waitcnt vmcnt(16)" ::: "memory");                                        \
            pending_stores = false;                                                                  \
        } else {                                                                                     \
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                         \
        }                                                                                            \
        STAMP(t2)                                                                                    \
        S_BARRIER();                                                                                 \
        STAMP(t3)                                                                                    \
        if (stamps) {                                                                                \
            acc_lds += t1 - t0;                                                                      \
            acc_vm += t2 - t1;                                                                       \
            acc_bar += t3 - t2;                                                                      \
            ++n_kt;                                                                                  \
        }                                                                                            \
        /* 7 */                                                                                      \
        if (!(LAST)) {                                                                               \
            READ_A(aP, NXT, 0, ch0)                                                                  \
            READ_B(bN0a, NXT, 0, ch0)                                                                \
            load_A(l2, CUR);                                                                         \
        }                                                                                            \
        MFMA8(aQ, bN1, 4, 2)                                                                         \
        PHASE_FENCE();                                                                               \
        /* 8 */                                                                                      \
        if (!(LAST)) {                                                                               \
            READ_B(bN1, NXT, 2, ch0)                                                                 \
            load_Blo(l2, CUR);                                                                       \
            load_Bhi(l2, CUR);                                                                       \
        }                                                                                            \
        MFMA8(aQ, bN0b, 4, 0)                                                                        \
        PHASE_FENCE();                                                                               \
        if (!(LAST)) advance(l2);                                                                    \
    }

    // diagnostic stamps (off unless a buffer is passed): where a K-tile's sync point spends its time
    unsigned long long t0 = 0, t1 = 0, t2 = 0, t3 = 0, acc_lds = 0, acc_vm = 0, acc_bar = 0, n_kt = 0, t_begin = 0, t_epi = 0, te0 = 0;
#define STAMP(var) if (stamps) { asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var) :: "memory"); }
    STAMP(t_begin)
    // nk is even (checked by the launcher), so every output tile starts in slot 0
    for (int ti = 0; ti < my_tiles; ++ti) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int kt = 0; kt + 2 < nk; kt += 2) {
            K_TILE(0, SLOT, false)
            K_TILE(SLOT, 0, false)
        }
        K_TILE(0, SLOT, false)
        K_TILE(SLOT, 0, true)
        STAMP(te0)
        epilogue(0, SLOT);  // K-tile s+2 of the stream goes to the slot of the last K-tile (slot 1)
        STAMP(t0)
        if (stamps) t_epi += t0 - te0;
        advance(l2);
        seek(cc, cc.tile + (int)gridDim.x);
        if (ti + 1 < my_tiles) {
            READ_A(aP, 0, 0, ch0)
            READ_B(bN0a, 0, 0, ch0)
            READ_B(bN1, 0, 2, ch0)
        }
    }
    if (stamps) {
        STAMP(t0)
        if (lane == 0) {
            unsigned long long* o = stamps + ((size_t)blockIdx.x * 8 + wave) * 8;
            o[0] = t0 - t_begin;
            o[1] = acc_lds;
            o[2] = acc_vm;
            o[3] = acc_bar;
            o[4] = t_epi;
            o[5] = n_kt;
        }
    }
#undef STAMP
#undef READ_A
#undef READ_B
#undef MFMA8
#undef K_TILE
}

template <int EPI>
hipError_t launch256s(const GemmArgs& g, hipStream_t s, unsigned long long* stamps) {
    static bool attr_set = false;
    const int smem = 8 * HALF;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm_bf16_tn_256s<EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    const int tiles_m = (g.M + TM - 1) / TM, tiles_n = (g.N + TN - 1) / TN;
    const int ntiles = tiles_m * tiles_n;
    const int grid = ntiles < 256 ? ntiles : 256;
    hipLaunchKernelGGL(gemm_bf16_tn_256s<EPI>, dim3(grid), dim3(512), smem, s, g, tiles_m, tiles_n, stamps);
    return hipGetLastError();
}

}  // namespace

hipError_t launch_gemm256s(int epilogue, const GemmArgs& g, hipStream_t s, unsigned long long* stamps) {
    if (g.M <= 0 || g.N <= 0) return hipSuccess;
    if (g.K <= 0 || (g.K % (2 * TK)) != 0) return hipErrorInvalidValue;  // even number of K-tiles
    switch (epilogue) {
        case EPI_BIAS: return launch256s<EPI_BIAS>(g, s, stamps);
        case EPI_BIAS_GELU: return launch256s<EPI_BIAS_GELU>(g, s, stamps);
        case EPI_BIAS_RES: return launch256s<EPI_BIAS_RES>(g, s, stamps);
        case EPI_PATCH: return launch256s<EPI_PATCH>(g, s, stamps);
        case EPI_F32: return launch256s<EPI_F32>(g, s, stamps);
        case EPI_LN_BIAS: return launch256s<EPI_LN_BIAS>(g, s, stamps);
        case EPI_LN_BIAS_GELU: return launch256s<EPI_LN_BIAS_GELU>(g, s, stamps);
        default: return hipErrorInvalidValue;
    }
}
