// bf16 "TN" GEMM on MFMA for gfx950:  C[M,N] = A[M,K] . W[N,K]^T  with fused epilogues.
//
// Serves K2 (patch embed), K4 (QKV), K6 (o_proj + residual), K7 (fc1 + erf-GELU, fc2 +
// residual) and K9 (all-pairs cosine, f32 out) of SURVEY.md §2a.  Both operands are
// K-contiguous (Hugging Face Linear weights are [out, in]; embeddings are row vectors), so
// an MFMA fragment is one 16-byte read on either side.
//
// Structure (128 x 128 x 64 tile, 4 waves as 2x2, each wave 64x64 = 4x4 MFMA 16x16x32):
//   * global -> LDS by LDS-DMA (global_load_lds_dwordx4): no VGPR round trip.  The LDS image
//     is lane-linear, so the bank-conflict swizzle is applied to the per-lane SOURCE address
//     and again on the ds_read (chunk ^= (row>>1)&7 within the 128-byte row).
//   * double-buffered LDS (2 x 32 KiB), one barrier per K-tile; tile k+1 streams in while
//     tile k feeds the matrix cores; 2 workgroups per CU cover each other's barriers.
//   * operands are fed "swapped" (W fragment as MFMA A, activation fragment as MFMA B) so the
//     accumulator holds 4 consecutive n per lane -> 8-byte bf16x4 / 16-byte f32x4 stores and
//     bias / residual vector loads.
//   * XCD-aware bijective tile order: consecutive tiles share the A row panel in one XCD's L2.
#include <cstdlib>

#include "common.h"
#include "gemm_epilogue.h"
#include "kernels.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = BM * BK * 2;  // 16 KiB per operand tile

template <int EPI>
__global__ __launch_bounds__(256, 2) void gemm_bf16_tn_128(GemmArgs g) {
    __shared__ __attribute__((aligned(16))) char lds[4 * TILE_BYTES];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tiles_n = (g.N + BN - 1) / BN, tiles_m = (g.M + BM - 1) / BM;
    const int id = xcd_remap(blockIdx.x, tiles_m * tiles_n);
    const int tm = id / tiles_n, tn = id - tm * tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    const size_t ldb = (size_t)g.K * 2;
    const char* Ag = (const char*)g.A;
    const char* Wg = (const char*)g.W;

    // Per-lane source offsets of the 4 LDS-DMA pieces this wave issues per operand tile.
    // Piece i covers tile rows 8i..8i+7 (1 KiB); lane -> (row, 16-byte slot); the slot holds
    // logical chunk slot ^ ((row>>1)&7).  Rows past the matrix edge re-read the last row.
    size_t a_src[4], w_src[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int r = (wave * 4 + t) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((r >> 1) & 7);
        const int ra = min(m0 + r, g.M - 1), rw = min(n0 + r, g.N - 1);
        a_src[t] = (size_t)ra * ldb + c * 16;
        w_src[t] = (size_t)rw * ldb + c * 16;
    }
    auto stage = [&](int buf, int kt) {
        char* la = lds + buf * 2 * TILE_BYTES;
        char* lw = la + TILE_BYTES;
        const size_t koff = (size_t)kt * (BK * 2);
#pragma unroll
        for (int t = 0; t < 4; ++t) glds16(Ag + a_src[t] + koff, la + (wave * 4 + t) * 1024);
#pragma unroll
        for (int t = 0; t < 4; ++t) glds16(Wg + w_src[t] + koff, lw + (wave * 4 + t) * 1024);
    };

    const int wr = wave >> 1, wc = wave & 1;
    const int fr = lane & 15, fq = lane >> 4;
    const int fsw = (lane >> 1) & 7;  // (row>>1)&7 of this lane's fragment rows
    int a_off[4], w_off[4];           // row byte offsets inside the tile
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        a_off[i] = (wr * 64 + i * 16 + fr) * (BK * 2);
        w_off[i] = (wc * 64 + i * 16 + fr) * (BK * 2);
    }

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk = g.K / BK;
    stage(0, 0);
    __syncthreads();
    int cur = 0;
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) stage(cur ^ 1, kt + 1);
        const char* la = lds + cur * 2 * TILE_BYTES;
        const char* lw = la + TILE_BYTES;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int ch = ((kk * 4 + fq) ^ fsw) << 4;
            bf16x8 fa[4], fw[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                fa[i] = *(const bf16x8*)(la + a_off[i] + ch);
                fw[i] = *(const bf16x8*)(lw + w_off[i] + ch);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[j], fa[i], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
        cur ^= 1;
    }

    // Epilogue.  acc[i][j][r]: m = m0 + wr*64 + i*16 + (lane&15), n = n0 + wc*64 + j*16 + (lane>>4)*4 + r
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + wr * 64 + i * 16 + fr;
        if (m >= g.M) continue;
        const EpiRow er = epi_row<EPI>(m);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + wc * 64 + j * 16 + fq * 4;
            if (n >= g.N) continue;
            epi_store<EPI>(g, m, er, n, acc[i][j]);
        }
    }
}

}  // namespace

hipError_t launch_gemm256r(int epilogue, const GemmArgs& g, hipStream_t s, int defer);

// variant: 0 = choose by shape, 1 = 128x128 tiles, 3 = 256x256 ping-pong kernel with a 3-deep activation ring (all 160 KiB
// of LDS), 4 / 6 / 5 = variant 3 with 4 / 6 / 8 of a lane's 16 stores deferred into the next tile's first K-tile (2 is
// accepted for old callers and means 3: the 2-slot-ring kernel it named was folded into gemm256r.hip)
static int resolve_variant(const GemmArgs& g, int variant) {
    if (variant == 0) {
        const int64_t tiles256 = (int64_t)((g.M + 255) / 256) * ((g.N + 255) / 256);
        // 128 tiles: measured on whole embed calls of 16..256 crops (tools/bench_small.py; MME_GEMM_MIN256 sweeps it): with the
        // threshold at 256, calls of 64 / 96 crops ran their 150- / 225-tile GEMMs on the 128 x 128 kernel and took 11 % / 9 %
        // longer (that kernel also leaves no LayerNorm partial sums: one more pass over x per LayerNorm)
        static const int min256 = diag_env("MME_GEMM_MIN256") ? atoi(diag_env("MME_GEMM_MIN256")) : 128;
        variant = tiles256 >= min256 ? 4 : 1;
    }
    if (variant == 2) variant = 3;
    if (g.K < 128) variant = 1;  // the 256 kernel streams two K-tiles ahead
    return variant;
}

bool gemm_runs_256(const GemmArgs& g, int variant) { return resolve_variant(g, variant) != 1; }

hipError_t launch_gemm(int epilogue, const GemmArgs& g, hipStream_t s, int variant) {
    if (g.M <= 0 || g.N <= 0) return hipSuccess;
    const bool automatic = variant == 0;
    variant = resolve_variant(g, variant);
    if (variant >= 3 && variant <= 6) {
        int defer = variant == 3 ? 0 : (variant == 4 ? 4 : (variant == 5 ? 8 : 6));
        // f32 out (K9): the deferred row block LOSES there -- interleaved A/B on the [8192 x 65536] block, 20 launches per arm,
        // four rounds: 0.911 ms without, 1.00 ms with (tools/k9_ab.py; the mid-round figure that favoured it had measured the
        // undeferred arm first, cold).  The bf16 epilogues keep it (-3.4 % of the forward's GEMM time, tools/ab_step.py).
        if (automatic && epilogue == EPI_F32) defer = 0;
        return launch_gemm256r(epilogue, g, s, defer);
    }
    if (g.K <= 0 || (g.K % BK) != 0) return hipErrorInvalidValue;
    const int tiles = ((g.M + BM - 1) / BM) * ((g.N + BN - 1) / BN);
    dim3 grid(tiles), block(256);
    switch (epilogue) {
        case EPI_BIAS: hipLaunchKernelGGL(gemm_bf16_tn_128<EPI_BIAS>, grid, block, 0, s, g); break;
        case EPI_BIAS_GELU: hipLaunchKernelGGL(gemm_bf16_tn_128<EPI_BIAS_GELU>, grid, block, 0, s, g); break;
        case EPI_BIAS_RES:
        case EPI_BIAS_RES_STATS:  // no partial planes from this kernel: the caller derives the statistics from x (gemm_runs_256)
            hipLaunchKernelGGL(gemm_bf16_tn_128<EPI_BIAS_RES>, grid, block, 0, s, g); break;
        case EPI_PATCH: hipLaunchKernelGGL(gemm_bf16_tn_128<EPI_PATCH>, grid, block, 0, s, g); break;
        case EPI_F32: hipLaunchKernelGGL(gemm_bf16_tn_128<EPI_F32>, grid, block, 0, s, g); break;
        case EPI_LN_BIAS: hipLaunchKernelGGL(gemm_bf16_tn_128<EPI_LN_BIAS>, grid, block, 0, s, g); break;
        case EPI_LN_BIAS_GELU: hipLaunchKernelGGL(gemm_bf16_tn_128<EPI_LN_BIAS_GELU>, grid, block, 0, s, g); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}
