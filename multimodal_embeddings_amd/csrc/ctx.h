// Private to the library: the context behind `mme_ctx*` and the helpers the C-ABI translation units share.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/mme.h"
#include "common.h"
#include "kernels.h"

struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
};

struct LayerDev {
    float *ln1_g, *ln1_b, *ln2_g, *ln2_b;
    bf16_t *qkv_w, *o_w, *fc1_w, *fc2_w;
    float *qkv_b, *o_b, *fc1_b, *fc2_b;
    // LayerNorm folded into the consuming GEMM: W' = bf16(W * gamma), colsum = sum_k W', b' = b + W . beta
    bf16_t *qkv_wf, *fc1_wf;
    float *qkv_cs, *qkv_bf, *fc1_cs, *fc1_bf;
};

enum KClass { KC_PRE = 0, KC_GEMM = 1, KC_LN = 2, KC_ATTN = 3, KC_POOL = 4, KC_COS = 5, KC_PAGE = 6, KC_CLUSTER = 7, KC_NEIGH = 8, KC_COMM = 9 };

struct EventPair {
    hipEvent_t a, b;
    int cls;
};

struct TileVitDev;  // capi_tilevit.hip

struct mme_ctx {
    int device = 0;
    std::string err;
    bool loaded = false;
    float ln_eps = 1e-12f;
    int chunk = 4096;
    int gemm_variant = 0;
    int ln_mode = 2;  // 0 LayerNorm kernel, 1 folded into the GEMMs + one statistics pass over x, 2 folded + partial sums from the producing epilogue
    int neigh_mode = 0;  // K12: 0 by size, 1 cosine block through the workspace, 2 fused candidate lists
    // weights
    std::vector<void*> allocs;
    float *cls = nullptr, *pos = nullptr, *patch_b = nullptr, *lnf_g = nullptr, *lnf_b = nullptr;
    bf16_t* patch_w = nullptr;
    LayerDev layer[VIT_L];
    float* lut = nullptr;  // [3,256]
    NormAffine norm_aff{};  // the same mapping as one fma per value where that is bit-exact after the bf16 rounding (set_lut)
    // workspace (sized for `chunk` crops)
    int ws_chunk = 0;
    DevBuf attn_guard;      // int[64]: one guard word per layer of a pass (attention.hip, FAST form)
    bool prune_last = false;  // mme_set_forward_pruning
    int zigzag = 1;           // forward_chunk: 1 = consecutive kernels walk the rows in opposite directions, 2 = attention only
    int attn_mode = 1;      // mme_set_attention_mode: 0 exact, 1 fast (guarded), 2 fast with the guard forced (tests)
    DevBuf x, hbuf, qkv, att, mlp, stats, lnpart, patches, tmp, htab, crops, hwork, page_ws, cluster_ws, neigh_ws, zero_bias;
    // host staging for crop tables
    std::vector<CropDesc> h_crops;
    std::vector<HWork> h_work;
    // profiling
    bool prof = false;
    std::vector<EventPair> events;
    size_t events_used = 0;
    // tile-ViT encoder option
    TileVitDev* tv = nullptr;
};

int fail(mme_ctx* c, int code, const char* fmt, ...);

#define HIP_TRY(c, expr)                                                                            \
    do {                                                                                            \
        hipError_t e_ = (expr);                                                                     \
        if (e_ != hipSuccess) return fail((c), MME_E_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

int ensure(mme_ctx* c, DevBuf& b, size_t bytes);
uint16_t f32_to_bf16_rne(float f);
int upload_f32(mme_ctx* c, const float* src, size_t n, float** dst);
// concatenates up to three [rows_i, cols] f32 matrices row-wise, converts to bf16, uploads; `scale` multiplies every value first
int upload_bf16(mme_ctx* c, const float* const* srcs, const size_t* rows, int nsrc, size_t cols, bf16_t** dst, float scale = 1.0f);
// LayerNorm folding for `y = W . LN(x) + b` (bs[i] may be null: no bias)
int upload_folded(mme_ctx* c, const float* const* ws, const float* const* bs, const size_t* rows, int nsrc, size_t cols, const float* gamma,
                  const float* beta, bf16_t** wf, float** cs, float** bf);
void tile_vit_free(mme_ctx* c);

struct Timed {
    mme_ctx* c;
    hipStream_t s;
    EventPair* ev = nullptr;
    Timed(mme_ctx* c_, hipStream_t s_, int cls);
    ~Timed();
};
