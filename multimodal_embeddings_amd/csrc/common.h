// Shared device helpers for the gfx950 (CDNA4) kernels.  64-wide wavefronts throughout.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) short s16x4;

// Experiment switches -- environment variables that change the tiling or (MME_GEMM_DEBUG, MME_ATTN_DEBUG) produce wrong
// results on purpose -- exist only in the diagnostic build (`python -m multimodal_embeddings_amd.build --diag` ->
// libmme_diag.so, -DMME_DIAG): a stray variable in a user's environment cannot change what libmme.so computes.
#include <stdlib.h>
#ifdef MME_DIAG
static inline const char* diag_env(const char* name) { return getenv(name); }
#else
static inline const char* diag_env(const char*) { return nullptr; }
#endif

#define GLOBAL_AS __attribute__((address_space(1)))
#define LDS_AS __attribute__((address_space(3)))

// ViT-B/16 @224 geometry (BASELINE.json north_star); fixed at compile time so every
// index computation folds to constants.
constexpr int VIT_D = 768;
constexpr int VIT_T = 197;
constexpr int VIT_NP = 196;
constexpr int VIT_H = 12;
constexpr int VIT_DH = 64;
constexpr int VIT_F = 3072;
constexpr int VIT_L = 12;
constexpr int VIT_GRID = 14;
constexpr int VIT_PATCH = 16;
constexpr int VIT_IMG = 224;

__device__ __forceinline__ float bf16_bits_to_f32(uint16_t b) { return __uint_as_float(((uint32_t)b) << 16); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// 16-byte asynchronous global -> LDS copy (LDS-DMA).  The LDS destination is the
// wave-uniform base + lane*16; only the SOURCE address is per lane.
__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)gsrc, (LDS_AS void*)lds_wave_base, 16, 0, 0);
}

// XCD-aware bijective block remap: consecutive logical tiles land on one XCD (its own
// L2), whatever the grid size.  `orig % 8` labels blocks sharing an XCD.
__device__ __forceinline__ int xcd_remap(int orig, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = orig & 7;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + (orig >> 3);
}
