// Row kernels of the tile-ViT encoder (the Mllama vision tower, transformers modeling_mllama.py MllamaVisionModel):
// im2col of the f32 tiles, assembly of the token sequence (class token, gated position / tile embeddings,
// layernorm_pre, zero padding rows), layernorm_post + post-tile embedding, and the output gather (final state +
// five intermediate states -> 7680 per token; class token of tile 0 L2-normalised as the crop's vector).
// All HBM-bound, one 64-lane wave per 1280-wide row (20 values per lane, f32 statistics, two-pass as torch).
#include "common.h"
#include "kernels.h"

namespace {

constexpr int D = 1280, TOK = 1601, TOKP = 1608, TILES = 4, GRID = 40, PS = 14, IMG = 560;
constexpr int PDIM = 588, PDIMP = 640;  // 3 * 14 * 14 patch elements, padded to a multiple of the GEMM's K step
constexpr int PER_LANE = D / 64;        // 20

// pixel_values f32 [n, 4, 3, 560, 560] -> patches bf16 [n * 4 * 1600, 640], element (c, ky, kx), zeros past 588
__global__ __launch_bounds__(256) void tile_patchify(const float* __restrict__ pv, bf16_t* __restrict__ patches, int64_t npatch) {
    const int64_t p = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (p >= npatch) return;
    const int lane = threadIdx.x & 63;
    const int64_t tile = p / (GRID * GRID);
    const int pp = (int)(p - tile * GRID * GRID), py = pp / GRID, px = pp - py * GRID;
    const float* src = pv + tile * 3 * IMG * IMG + (py * PS) * IMG + px * PS;
    bf16_t* dst = patches + p * PDIMP;
    for (int e = lane; e < PDIMP; e += 64) {
        float v = 0.f;
        if (e < PDIM) {
            const int c = e / (PS * PS), rem = e - c * PS * PS, ky = rem / PS, kx = rem - ky * PS;
            v = src[c * IMG * IMG + ky * IMG + kx];
        }
        dst[e] = (bf16_t)v;
    }
}

struct RowStats {
    float mean, rstd;
};
__device__ __forceinline__ RowStats row_stats(const float (&v)[PER_LANE], float eps) {
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < PER_LANE; ++j) s += v[j];
    const float mean = wave_sum(s) * (1.0f / D);
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < PER_LANE; ++j) {
        const float d = v[j] - mean;
        q += d * d;
    }
    return {mean, rsqrtf(wave_sum(q) * (1.0f / D) + eps)};
}

// x[(img, tile, tok)] = layernorm_pre(token) for tok < 1601, zeros for the 7 padding rows.
// token = class_emb (tok 0) | patch_emb[(img, tile, tok - 1)] + pre[aid][tile]   , + pos[tok] + tilepos[aid][tile][tok]
// (pre / pos / tilepos already carry their tanh gates, applied on the host at load time)
__global__ __launch_bounds__(256) void tile_assemble(const bf16_t* __restrict__ pemb, const float* __restrict__ cls, const float* __restrict__ pre,
                                                     const float* __restrict__ pos, const float* __restrict__ tilepos,
                                                     const float* __restrict__ g, const float* __restrict__ b, const int32_t* __restrict__ aid,
                                                     bf16_t* __restrict__ x, int64_t rows, float eps) {
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    const int64_t it = row / TOKP;  // image * 4 + tile
    const int tok = (int)(row - it * TOKP), tile = (int)(it & 3), img = (int)(it >> 2);
    bf16_t* xr = x + row * D;
    if (tok >= TOK) {
#pragma unroll
        for (int j = 0; j < PER_LANE; j += 4) *(uint2*)(xr + (j / 4) * 256 + lane * 4) = make_uint2(0, 0);
        return;
    }
    const int a = aid[img];
    float v[PER_LANE];
#pragma unroll
    for (int j = 0; j < PER_LANE; j += 4) {
        const int c = (j / 4) * 256 + lane * 4;
        f32x4 t;
        if (tok == 0) {
            t = *(const f32x4*)(cls + c);
        } else {
            const bf16x4 pe = *(const bf16x4*)(pemb + (it * (GRID * GRID) + tok - 1) * D + c);
            const f32x4 pr = *(const f32x4*)(pre + ((int64_t)a * TILES + tile) * D + c);
            t = f32x4{(float)pe[0] + pr[0], (float)pe[1] + pr[1], (float)pe[2] + pr[2], (float)pe[3] + pr[3]};
        }
        const f32x4 po = *(const f32x4*)(pos + (int64_t)tok * D + c);
        const f32x4 tp = *(const f32x4*)(tilepos + (((int64_t)a * TILES + tile) * TOK + tok) * D + c);
#pragma unroll
        for (int k = 0; k < 4; ++k) v[j + k] = t[k] + po[k] + tp[k];
    }
    const RowStats st = row_stats(v, eps);
#pragma unroll
    for (int j = 0; j < PER_LANE; j += 4) {
        const int c = (j / 4) * 256 + lane * 4;
        const f32x4 gv = *(const f32x4*)(g + c), bv = *(const f32x4*)(b + c);
        bf16x4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = (bf16_t)((v[j + k] - st.mean) * st.rstd * gv[k] + bv[k]);
        *(bf16x4*)(xr + c) = o;
    }
}

// x <- layernorm_post(x) + post[aid][tile], every row of the padded sequence (the reference normalises and
// shifts the padding rows as well; they stay keys and values of the global layers)
__global__ __launch_bounds__(256) void tile_ln_post(bf16_t* __restrict__ x, const float* __restrict__ g, const float* __restrict__ b,
                                                    const float* __restrict__ post, const int32_t* __restrict__ aid, int64_t rows, float eps) {
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    const int64_t it = row / TOKP;
    const int tile = (int)(it & 3), a = aid[it >> 2];
    bf16_t* xr = x + row * D;
    float v[PER_LANE];
#pragma unroll
    for (int j = 0; j < PER_LANE; j += 4) {
        const bf16x4 p = *(const bf16x4*)(xr + (j / 4) * 256 + lane * 4);
#pragma unroll
        for (int k = 0; k < 4; ++k) v[j + k] = (float)p[k];
    }
    const RowStats st = row_stats(v, eps);
#pragma unroll
    for (int j = 0; j < PER_LANE; j += 4) {
        const int c = (j / 4) * 256 + lane * 4;
        const f32x4 gv = *(const f32x4*)(g + c), bv = *(const f32x4*)(b + c);
        const f32x4 po = *(const f32x4*)(post + ((int64_t)a * TILES + tile) * D + c);
        bf16x4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = (bf16_t)((v[j + k] - st.mean) * st.rstd * gv[k] + bv[k] + po[k]);
        *(bf16x4*)(xr + c) = o;
    }
}

// last_hidden_state f32 [n, 4, 1601, 1280 * (1 + ni)]: features [0, 1280) = final state, 1280 + d * ni + k = state k
// of the saved intermediate layers at dimension d (torch.stack(..., dim=-1) then flatten); padding rows dropped
__global__ __launch_bounds__(256) void tile_output(const bf16_t* __restrict__ x, const bf16_t* __restrict__ inter, int ni, int64_t inter_stride,
                                                   float* __restrict__ hidden, int64_t out_rows) {
    const int64_t orow = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (orow >= out_rows) return;
    const int lane = threadIdx.x & 63;
    const int64_t it = orow / TOK;
    const int tok = (int)(orow - it * TOK);
    const int64_t row = it * TOKP + tok;
    const int F = D * (1 + ni);
    float* o = hidden + orow * F;
    for (int c = lane; c < D; c += 64) o[c] = (float)x[row * D + c];
    for (int e = lane; e < D * ni; e += 64) {
        const int d = e / ni, k = e - d * ni;
        o[D + e] = (float)inter[k * inter_stride + row * D + d];
    }
}

// the crop's vector: class token (token 0) of tile 0, all 1280 * (1 + ni) features, L2-normalised (the pooling rule of
// deprecated_package/embedder.py:17-34: one token row, F.normalize); one workgroup per image
__global__ __launch_bounds__(256) void tile_pool(const bf16_t* __restrict__ x, const bf16_t* __restrict__ inter, int ni, int64_t inter_stride,
                                                 float* __restrict__ emb_f32, bf16_t* __restrict__ emb_bf16) {
    __shared__ float part[4];
    const int img = blockIdx.x, tid = threadIdx.x;
    const int64_t row = (int64_t)img * TILES * TOKP;
    const int F = D * (1 + ni);
    float n2 = 0.f;
    for (int e = tid; e < F; e += 256) {
        float v;
        if (e < D) {
            v = (float)x[row * D + e];
        } else {
            const int d = (e - D) / ni, k = (e - D) - d * ni;
            v = (float)inter[k * inter_stride + row * D + d];
        }
        n2 += v * v;
    }
    n2 = wave_sum(n2);
    if ((tid & 63) == 0) part[tid >> 6] = n2;
    __syncthreads();
    const float inv = 1.0f / fmaxf(sqrtf(part[0] + part[1] + part[2] + part[3]), 1e-12f);
    for (int e = tid; e < F; e += 256) {
        float v;
        if (e < D) {
            v = (float)x[row * D + e];
        } else {
            const int d = (e - D) / ni, k = (e - D) - d * ni;
            v = (float)inter[k * inter_stride + row * D + d];
        }
        if (emb_f32) emb_f32[(int64_t)img * F + e] = v * inv;
        if (emb_bf16) emb_bf16[(int64_t)img * F + e] = (bf16_t)(v * inv);
    }
}

}  // namespace

hipError_t launch_tile_patchify(const float* pv, void* patches, int64_t npatch, hipStream_t s) {
    if (npatch <= 0) return hipSuccess;
    hipLaunchKernelGGL(tile_patchify, dim3((unsigned)((npatch + 3) / 4)), dim3(256), 0, s, pv, (bf16_t*)patches, npatch);
    return hipGetLastError();
}
hipError_t launch_tile_assemble(const void* pemb, const float* cls, const float* pre, const float* pos, const float* tilepos, const float* g,
                                const float* b, const int32_t* aid, void* x, int64_t rows, float eps, hipStream_t s) {
    if (rows <= 0) return hipSuccess;
    hipLaunchKernelGGL(tile_assemble, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, (const bf16_t*)pemb, cls, pre, pos, tilepos, g, b, aid,
                       (bf16_t*)x, rows, eps);
    return hipGetLastError();
}
hipError_t launch_tile_ln_post(void* x, const float* g, const float* b, const float* post, const int32_t* aid, int64_t rows, float eps, hipStream_t s) {
    if (rows <= 0) return hipSuccess;
    hipLaunchKernelGGL(tile_ln_post, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, (bf16_t*)x, g, b, post, aid, rows, eps);
    return hipGetLastError();
}
hipError_t launch_tile_output(const void* x, const void* inter, int ni, int64_t inter_stride, float* hidden, int64_t out_rows, hipStream_t s) {
    if (out_rows <= 0) return hipSuccess;
    hipLaunchKernelGGL(tile_output, dim3((unsigned)((out_rows + 3) / 4)), dim3(256), 0, s, (const bf16_t*)x, (const bf16_t*)inter, ni, inter_stride,
                       hidden, out_rows);
    return hipGetLastError();
}
hipError_t launch_tile_pool(const void* x, const void* inter, int ni, int64_t inter_stride, int n, float* emb_f32, void* emb_bf16, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(tile_pool, dim3(n), dim3(256), 0, s, (const bf16_t*)x, (const bf16_t*)inter, ni, inter_stride, emb_f32, (bf16_t*)emb_bf16);
    return hipGetLastError();
}
