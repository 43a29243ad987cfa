// Row-wise HBM-bound kernels of the ViT forward: LayerNorm (K3), [CLS] row initialisation,
// and final-LayerNorm + pooling + L2 normalisation (K8).
//
// One 64-lane wave owns one 768-wide row: 3 x 8-byte (bf16x4) loads per lane, statistics
// in f32 registers, two wave reductions (mean, then centred variance -- the same two-pass
// form torch's LayerNorm uses, transformers modeling_vit.py:261-262,348), 8-byte stores.
#include "common.h"
#include "gemm_epilogue.h"
#include "kernels.h"

namespace {

__global__ __launch_bounds__(256) void layernorm_rows(const bf16_t* __restrict__ x, const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, bf16_t* __restrict__ y,
                                                      int64_t rows, float eps) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const bf16_t* xr = x + row * VIT_D;
    float v[12];
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        const bf16x4 p = *(const bf16x4*)(xr + t * 256 + lane * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[t * 4 + j] = (float)p[j];
    }
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 12; ++j) s += v[j];
    const float mean = wave_sum(s) * (1.0f / VIT_D);
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < 12; ++j) {
        v[j] -= mean;
        q += v[j] * v[j];
    }
    const float rstd = rsqrtf(wave_sum(q) * (1.0f / VIT_D) + eps);
    bf16_t* yr = y + row * VIT_D;
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        const int c = t * 256 + lane * 4;
        const f32x4 gv = *(const f32x4*)(gamma + c);
        const f32x4 bv = *(const f32x4*)(beta + c);
        bf16x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = (bf16_t)(v[t * 4 + j] * rstd * gv[j] + bv[j]);
        *(bf16x4*)(yr + c) = o;
    }
}

// LayerNorm statistics only (mean, rstd) of bf16 rows of 768: the normalisation itself is
// folded into the GEMM that consumes the row (EPI_LN_*), so the residual stream is read once
// and nothing is written back but 8 bytes per row.  Same two-pass f32 arithmetic as
// layernorm_rows, so the folded path sees the statistics LayerNorm would have used.
__global__ __launch_bounds__(256) void ln_stats_rows(const bf16_t* __restrict__ x, int64_t rows, float eps, float* __restrict__ stats) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const bf16_t* xr = x + row * VIT_D;
    float v[12];
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        const bf16x4 p = *(const bf16x4*)(xr + t * 256 + lane * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[t * 4 + j] = (float)p[j];
    }
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 12; ++j) s += v[j];
    const float mean = wave_sum(s) * (1.0f / VIT_D);
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < 12; ++j) {
        const float d = v[j] - mean;
        q += d * d;
    }
    const float rstd = rsqrtf(wave_sum(q) * (1.0f / VIT_D) + eps);
    if (lane == 0) *(float2*)(stats + 2 * row) = make_float2(mean, rstd);
}

// The same statistics in the canonical order of gemm_epilogue.h (ln_accumulate / ln_finish_row): one wave per
// row, lane = (slice, column group g) for up to two rounds of 16 slices; 48 of the 64 lanes carry data at d = 768.  Used wherever no
// EPI_BIAS_RES_STATS epilogue produced the partial sums: the first LayerNorm of a pass, small problems that run
// the 128 x 128 kernel, the rows of a ragged last row tile.
__global__ __launch_bounds__(256) void ln_stats_canonical_rows(const bf16_t* __restrict__ x, int64_t row0, int64_t row1, int d, float eps,
                                                               float* __restrict__ stats, int64_t stride) {
    const int lane = threadIdx.x & 63;
    const int64_t row = row0 + ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * stride;  // rows row0, row0 + stride, ... < row1
    if (row >= row1) return;
    const int slice = lane >> 2, grp = lane & 3, nslice = d >> 6;  // a lane serves slices `slice` and `slice + 16` (d <= 2048)
    float s[2] = {0.f, 0.f}, q[2] = {0.f, 0.f};
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int sl = slice + 16 * h;
        if (sl < nslice) {
            const bf16_t* xr = x + row * d + sl * 64 + grp * 4;
            uint2 pk[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) pk[j] = *(const uint2*)(xr + j * 16);
#pragma unroll
            for (int j = 0; j < 4; ++j) ln_accumulate(pk[j], s[h], q[h]);
        }
        s[h] += __shfl_xor(s[h], 1, 64);  // g0 + g1 | g2 + g3
        q[h] += __shfl_xor(q[h], 1, 64);
        s[h] += __shfl_xor(s[h], 2, 64);  // (g0 + g1) + (g2 + g3)
        q[h] += __shfl_xor(q[h], 2, 64);
    }
    double S = 0.0, Q = 0.0;
    for (int k = 0; k < nslice; ++k) {
        S += (double)__shfl(k < 16 ? s[0] : s[1], 4 * (k & 15), 64);
        Q += (double)__shfl(k < 16 ? q[0] : q[1], 4 * (k & 15), 64);
    }
    if (lane == 0) *(float2*)(stats + 2 * row) = ln_finish_row(S, Q, d, eps);
}

// rows [0, rows): the partial planes of an EPI_BIAS_RES_STATS GEMM -> (mean, rstd); one thread per row, the
// plane reads are coalesced over the rows
__global__ __launch_bounds__(256) void ln_finish_rows(const float* __restrict__ part, int64_t part_rows, int64_t rows, int d, float eps,
                                                      float* __restrict__ stats) {
    const int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (row >= rows) return;
    const int nslice = d >> 6;
    double S = 0.0, Q = 0.0;
    for (int k = 0; k < nslice; ++k) {
        S += (double)part[(int64_t)k * part_rows + row];
        Q += (double)part[(int64_t)(nslice + k) * part_rows + row];
    }
    *(float2*)(stats + 2 * row) = ln_finish_row(S, Q, d, eps);
}

__global__ __launch_bounds__(256) void cls_rows(bf16_t* __restrict__ x, const float* __restrict__ cls,
                                                const float* __restrict__ pos, int B) {
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= B) return;
    bf16_t* xr = x + (int64_t)b * VIT_T * VIT_D;
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        const int c = t * 256 + lane * 4;
        const f32x4 a = *(const f32x4*)(cls + c);
        const f32x4 p = *(const f32x4*)(pos + c);
        bf16x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = (bf16_t)(a[j] + p[j]);
        *(bf16x4*)(xr + c) = o;
    }
}

// K8: restates last_pooling (deprecated_package/embedder.py:17-34) for one fixed token
// index per sequence, after the final LayerNorm of that row only (the other 196 rows of
// the last hidden state are never read by the reference's pooling).
__global__ __launch_bounds__(256) void pool_ln_l2(const bf16_t* __restrict__ x, const float* __restrict__ gamma,
                                                  const float* __restrict__ beta, int B, int tok, float eps,
                                                  float* __restrict__ emb_f32, bf16_t* __restrict__ emb_bf16) {
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= B) return;
    const bf16_t* xr = x + ((int64_t)b * VIT_T + tok) * VIT_D;
    float v[12];
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        const bf16x4 p = *(const bf16x4*)(xr + t * 256 + lane * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[t * 4 + j] = (float)p[j];
    }
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 12; ++j) s += v[j];
    const float mean = wave_sum(s) * (1.0f / VIT_D);
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < 12; ++j) {
        v[j] -= mean;
        q += v[j] * v[j];
    }
    const float rstd = rsqrtf(wave_sum(q) * (1.0f / VIT_D) + eps);
    float n2 = 0.f;
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        const int c = t * 256 + lane * 4;
        const f32x4 gv = *(const f32x4*)(gamma + c);
        const f32x4 bv = *(const f32x4*)(beta + c);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            v[t * 4 + j] = v[t * 4 + j] * rstd * gv[j] + bv[j];
            n2 += v[t * 4 + j] * v[t * 4 + j];
        }
    }
    // torch.nn.functional.normalize: x / max(||x||_2, 1e-12)
    const float inv = 1.0f / fmaxf(sqrtf(wave_sum(n2)), 1e-12f);
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        const int c = t * 256 + lane * 4;
        f32x4 o;
        bf16x4 ob;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            o[j] = v[t * 4 + j] * inv;
            ob[j] = (bf16_t)o[j];
        }
        if (emb_f32) *(f32x4*)(emb_f32 + (int64_t)b * VIT_D + c) = o;
        if (emb_bf16) *(bf16x4*)(emb_bf16 + (int64_t)b * VIT_D + c) = ob;
    }
}

// f32 rows of any width d (d % 4 == 0) -> L2-normalised bf16 rows (one wave per row).  The
// reference hands vectors around as Python float lists (embedder.py:132); this is the way
// such vectors enter the bf16 cosine kernel, with the same normalisation as last_pooling.
__global__ __launch_bounds__(256) void normalise_rows_f32(const float* __restrict__ x, int64_t rows, int d, bf16_t* __restrict__ y) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + row * d;
    float n2 = 0.f;
    for (int c = lane * 4; c < d; c += 256) {
        const f32x4 v = *(const f32x4*)(xr + c);
        n2 += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
    }
    const float inv = 1.0f / fmaxf(sqrtf(wave_sum(n2)), 1e-12f);
    bf16_t* yr = y + row * d;
    for (int c = lane * 4; c < d; c += 256) {
        const f32x4 v = *(const f32x4*)(xr + c);
        bf16x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = (bf16_t)(v[j] * inv);
        *(bf16x4*)(yr + c) = o;
    }
}

}  // namespace

hipError_t launch_normalise_rows(const float* x, int64_t rows, int d, void* y, hipStream_t s) {
    if (rows <= 0) return hipSuccess;
    hipLaunchKernelGGL(normalise_rows_f32, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, x, rows, d, (bf16_t*)y);
    return hipGetLastError();
}

hipError_t launch_layernorm(const void* x, const float* gamma, const float* beta, void* y, int64_t rows, float eps, hipStream_t s) {
    if (rows <= 0) return hipSuccess;
    hipLaunchKernelGGL(layernorm_rows, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, (const bf16_t*)x, gamma, beta,
                       (bf16_t*)y, rows, eps);
    return hipGetLastError();
}

hipError_t launch_ln_stats(const void* x, int64_t rows, float eps, float* stats, hipStream_t s) {
    if (rows <= 0) return hipSuccess;
    hipLaunchKernelGGL(ln_stats_rows, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, (const bf16_t*)x, rows, eps, stats);
    return hipGetLastError();
}

hipError_t launch_ln_stats_canonical(const void* x, int64_t row0, int64_t row1, int d, float eps, float* stats, hipStream_t s, int64_t stride) {
    if (row1 <= row0) return hipSuccess;
    if (d <= 0 || (d % 64) != 0 || d > 2048 || stride < 1) return hipErrorInvalidValue;
    const int64_t nrows = (row1 - row0 + stride - 1) / stride;
    hipLaunchKernelGGL(ln_stats_canonical_rows, dim3((unsigned)((nrows + 3) / 4)), dim3(256), 0, s, (const bf16_t*)x, row0, row1, d, eps,
                       stats, stride);
    return hipGetLastError();
}

hipError_t launch_ln_finish(const float* part, int64_t part_rows, int64_t rows, int d, float eps, float* stats, hipStream_t s) {
    if (rows <= 0) return hipSuccess;
    hipLaunchKernelGGL(ln_finish_rows, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, s, part, part_rows, rows, d, eps, stats);
    return hipGetLastError();
}

hipError_t launch_cls_rows(void* x, const float* cls, const float* pos, int B, hipStream_t s) {
    if (B <= 0) return hipSuccess;
    hipLaunchKernelGGL(cls_rows, dim3((B + 3) / 4), dim3(256), 0, s, (bf16_t*)x, cls, pos, B);
    return hipGetLastError();
}

hipError_t launch_pool(const void* x, const float* gamma, const float* beta, int B, int tok, float eps, float* emb_f32,
                       void* emb_bf16, hipStream_t s) {
    if (B <= 0) return hipSuccess;
    hipLaunchKernelGGL(pool_ln_l2, dim3((B + 3) / 4), dim3(256), 0, s, (const bf16_t*)x, gamma, beta, B, tok, eps, emb_f32,
                       (bf16_t*)emb_bf16);
    return hipGetLastError();
}
