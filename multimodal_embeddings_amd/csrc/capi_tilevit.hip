// C ABI of the tile-ViT encoder option: the reference encoder's own vision tower geometry
// (deprecated_package/config.py:58 -> transformers MllamaVisionModel: <= 4 tiles of 560 x 560, patch 14, 1601 tokens per
// tile, 1280-d, 16 heads, 32 local + 8 gated global layers, 7680-d output) on the same MFMA GEMM, LayerNorm-folding
// and statistics machinery as the ViT-B/16 path, with its own flash-style attention kernel (attention_tiles.hip).
#include <cmath>
#include <cstring>
#include <new>
#include <vector>

#include "ctx.h"

namespace {
constexpr int TD = 1280, TF = 5120, TH = 16, TTOK = 1601, TTOKP = 1608, TTILES = 4, TT = TTILES * TTOKP, TGRID = 40;
constexpr int TPDIM = 588, TPDIMP = 640, TMAXI = 8, TARATIOS = 9;
}  // namespace

struct TileLayerDev {
    bf16_t *qkv_wf, *o_w, *fc1_wf, *fc2_w;
    float *qkv_cs, *qkv_bf, *fc1_cs, *fc1_bf, *fc2_b;
};

struct TileVitDev {
    int layers = 0, global_layers = 0, ni = 0;
    int inter_after[TMAXI];
    int save_before = 0;  // mme_tile_vit_weights.intermediate_save_point
    float eps = 1e-5f;
    float *cls = nullptr, *pre = nullptr, *pos = nullptr, *tilepos = nullptr, *post = nullptr;
    float *lnpre_g = nullptr, *lnpre_b = nullptr, *lnpost_g = nullptr, *lnpost_b = nullptr, *zeros = nullptr;
    bf16_t* patch_w = nullptr;
    std::vector<TileLayerDev> layer;
    std::vector<void*> allocs;
    // workspace for `ws_images` images
    int ws_images = 0;
    DevBuf patches, pemb, x, qkv, att, mlp, stats, lnpart, inter, meta;
};

void tile_vit_free(mme_ctx* c) {
    if (!c->tv) return;
    TileVitDev* t = c->tv;
    for (void* p : t->allocs) (void)hipFree(p);
    DevBuf* bufs[] = {&t->patches, &t->pemb, &t->x, &t->qkv, &t->att, &t->mlp, &t->stats, &t->lnpart, &t->inter, &t->meta};
    for (DevBuf* b : bufs)
        if (b->p) (void)hipFree(b->p);
    delete t;
    c->tv = nullptr;
}

namespace {

// upload helpers of capi.hip register their allocations in c->allocs (freed by mme_destroy); fine for these too
int upload_scaled_f32(mme_ctx* c, const float* src, size_t n, float scale, float** dst) {
    std::vector<float> h(n);
    for (size_t i = 0; i < n; ++i) h[i] = src[i] * scale;
    return upload_f32(c, h.data(), n, dst);
}

}  // namespace

extern "C" {

int mme_load_tile_vit(mme_ctx* c, const mme_tile_vit_weights* w) {
    if (!c || !w) return fail(c, MME_E_ARG, "mme_load_tile_vit: null argument");
    if (w->image_size != 560 || w->patch_size != 14 || w->hidden != TD || w->heads != TH || w->mlp != TF || w->max_tiles != TTILES ||
        w->aspect_ratios != TARATIOS)
        return fail(c, MME_E_ARG, "mme_load_tile_vit: only the Mllama vision geometry 560/14/1280/16/5120, 4 tiles, 9 aspect-ratio rows is built; got %d/%d/%d/%d/%d, %d tiles, %d rows",
                    w->image_size, w->patch_size, w->hidden, w->heads, w->mlp, w->max_tiles, w->aspect_ratios);
    if (w->layers < 1 || w->global_layers < 0 || w->layers + w->global_layers > 256 || w->n_intermediate < 0 || w->n_intermediate > TMAXI)
        return fail(c, MME_E_ARG, "mme_load_tile_vit: bad layer counts (%d local, %d global, %d intermediate)", w->layers, w->global_layers, w->n_intermediate);
    for (int k = 0; k < w->n_intermediate; ++k)
        if (w->intermediate[k] < 0 || w->intermediate[k] >= w->layers || (k && w->intermediate[k] <= w->intermediate[k - 1]))
            return fail(c, MME_E_ARG, "mme_load_tile_vit: intermediate layer indices must be ascending and inside the local stack");
    if (w->intermediate_save_point != MME_TILE_SAVE_AFTER_LAYER && w->intermediate_save_point != MME_TILE_SAVE_BEFORE_LAYER)
        return fail(c, MME_E_ARG, "mme_load_tile_vit: intermediate_save_point must be MME_TILE_SAVE_AFTER_LAYER (0) or MME_TILE_SAVE_BEFORE_LAYER (1), got %d",
                    w->intermediate_save_point);
    if (!w->class_embedding || !w->patch_w || !w->pos_emb || !w->tile_pos_emb || !w->pre_emb || !w->post_emb || !w->ln_pre_g || !w->ln_pre_b ||
        !w->ln_post_g || !w->ln_post_b || !w->layer)
        return fail(c, MME_E_ARG, "mme_load_tile_vit: null tensor pointer");
    if (c->tv) return fail(c, MME_E_STATE, "mme_load_tile_vit: tile-ViT weights already loaded; create a new context");
    HIP_TRY(c, hipSetDevice(c->device));
    TileVitDev* t = new (std::nothrow) TileVitDev();
    if (!t) return fail(c, MME_E_NOMEM, "mme_load_tile_vit: out of host memory");
    c->tv = t;
    t->layers = w->layers;
    t->global_layers = w->global_layers;
    t->ni = w->n_intermediate;
    for (int k = 0; k < t->ni; ++k) t->inter_after[k] = w->intermediate[k];
    t->save_before = w->intermediate_save_point;
    t->eps = w->norm_eps;
    int r;
    // gates are applied here, once: the kernels add plain tables
    const float g_pos = std::tanh(w->pos_gate), g_pre = std::tanh(w->pre_gate), g_post = std::tanh(w->post_gate);
    if ((r = upload_f32(c, w->class_embedding, TD, &t->cls))) return r;
    if ((r = upload_scaled_f32(c, w->pos_emb, (size_t)TTOK * TD, 1.0f - g_pos, &t->pos))) return r;
    if ((r = upload_scaled_f32(c, w->tile_pos_emb, (size_t)TARATIOS * TTILES * TTOK * TD, g_pos, &t->tilepos))) return r;
    if ((r = upload_scaled_f32(c, w->pre_emb, (size_t)TARATIOS * TTILES * TD, g_pre, &t->pre))) return r;
    if ((r = upload_scaled_f32(c, w->post_emb, (size_t)TARATIOS * TTILES * TD, g_post, &t->post))) return r;
    if ((r = upload_f32(c, w->ln_pre_g, TD, &t->lnpre_g))) return r;
    if ((r = upload_f32(c, w->ln_pre_b, TD, &t->lnpre_b))) return r;
    if ((r = upload_f32(c, w->ln_post_g, TD, &t->lnpost_g))) return r;
    if ((r = upload_f32(c, w->ln_post_b, TD, &t->lnpost_b))) return r;
    {
        std::vector<float> z(TF, 0.f);
        if ((r = upload_f32(c, z.data(), TF, &t->zeros))) return r;
        // patch projection [1280, 588] -> [1280, 640] (zero columns: the GEMM's K step is 64)
        std::vector<float> pw((size_t)TD * TPDIMP, 0.f);
        for (int n = 0; n < TD; ++n) memcpy(&pw[(size_t)n * TPDIMP], w->patch_w + (size_t)n * TPDIM, TPDIM * sizeof(float));
        const float* src[1] = {pw.data()};
        const size_t rows[1] = {TD};
        if ((r = upload_bf16(c, src, rows, 1, TPDIMP, &t->patch_w))) return r;
    }
    const int L = w->layers + w->global_layers;
    t->layer.resize(L);
    for (int l = 0; l < L; ++l) {
        const mme_tile_layer& a = w->layer[l];
        const float* all[] = {a.ln1_g, a.ln1_b, a.q_w, a.k_w, a.v_w, a.o_w, a.ln2_g, a.ln2_b, a.fc1_w, a.fc1_b, a.fc2_w, a.fc2_b};
        for (const float* p : all)
            if (!p) return fail(c, MME_E_ARG, "mme_load_tile_vit: layer %d has a null tensor pointer", l);
        TileLayerDev& Ld = t->layer[l];
        // x + tanh(gate) * branch(x): the gate multiplies the branch's LAST linear map (global layers only)
        const float ga = a.gated ? std::tanh(a.gate_attn) : 1.0f, gf = a.gated ? std::tanh(a.gate_ffn) : 1.0f;
        // The attention kernels take their scores in log2 units straight from the matrix pipe (attention_tiles.hip): 80^-0.5 * log2(e)
        // is folded into the query projection here, once, BEFORE the rounding to bf16 that the upload applies anyway (as mme_load_vit does)
        const float qsc = 0.11180339887498949f * 1.44269504088896341f;
        std::vector<float> qw_s((size_t)TD * TD);
        for (size_t i = 0; i < qw_s.size(); ++i) qw_s[i] = a.q_w[i] * qsc;
        const float* qkv[3] = {qw_s.data(), a.k_w, a.v_w};
        const float* nob[3] = {nullptr, nullptr, nullptr};
        const size_t r3[3] = {TD, TD, TD};
        if ((r = upload_folded(c, qkv, nob, r3, 3, TD, a.ln1_g, a.ln1_b, &Ld.qkv_wf, &Ld.qkv_cs, &Ld.qkv_bf))) return r;
        const float* o[1] = {a.o_w};
        const size_t r1[1] = {TD};
        if ((r = upload_bf16(c, o, r1, 1, TD, &Ld.o_w, ga))) return r;
        const float* f1[1] = {a.fc1_w};
        const float* f1b[1] = {a.fc1_b};
        const size_t rf[1] = {TF};
        if ((r = upload_folded(c, f1, f1b, rf, 1, TD, a.ln2_g, a.ln2_b, &Ld.fc1_wf, &Ld.fc1_cs, &Ld.fc1_bf))) return r;
        const float* f2[1] = {a.fc2_w};
        if ((r = upload_bf16(c, f2, r1, 1, TF, &Ld.fc2_w, gf))) return r;
        if ((r = upload_scaled_f32(c, a.fc2_b, TD, gf, &Ld.fc2_b))) return r;
    }
    return MME_OK;
}

int mme_tile_vit_forward(mme_ctx* c, const float* pixel_values, const int32_t* aspect_ids_host, const int32_t* num_tiles_host, int n,
                         float* hidden, float* emb_f32, uint16_t* emb_bf16, void* stream) {
    if (!c) return MME_E_ARG;
    if (!c->tv) return fail(c, MME_E_STATE, "mme_tile_vit_forward: call mme_load_tile_vit first");
    if (n < 0 || (n > 0 && (!pixel_values || !aspect_ids_host || !num_tiles_host))) return fail(c, MME_E_ARG, "mme_tile_vit_forward: null argument or n<0");
    if (n == 0) return MME_OK;
    for (int i = 0; i < n; ++i)
        if (aspect_ids_host[i] < 1 || aspect_ids_host[i] >= TARATIOS || num_tiles_host[i] < 1 || num_tiles_host[i] > TTILES)
            return fail(c, MME_E_ARG, "mme_tile_vit_forward: image %d has aspect-ratio id %d / %d tiles (ids 1..8, tiles 1..4)", i, aspect_ids_host[i], num_tiles_host[i]);
    HIP_TRY(c, hipSetDevice(c->device));
    TileVitDev* t = c->tv;
    hipStream_t s = (hipStream_t)stream;
    // images per pass: the workspace is ~230 MB per image (qkv 49, mlp 66, five intermediate states 82, ...)
    const int per_pass = c->chunk >= 64 ? 64 : (c->chunk < 1 ? 1 : c->chunk);
    const int cap = n < per_pass ? n : per_pass;
    if (t->ws_images < cap) {
        const size_t rows = (size_t)cap * TT;
        int r;
        if ((r = ensure(c, t->patches, (size_t)cap * TTILES * TGRID * TGRID * TPDIMP * 2))) return r;
        if ((r = ensure(c, t->pemb, (size_t)cap * TTILES * TGRID * TGRID * TD * 2))) return r;
        if ((r = ensure(c, t->x, rows * TD * 2))) return r;
        if ((r = ensure(c, t->qkv, rows * 3 * TD * 2))) return r;
        if ((r = ensure(c, t->att, rows * TD * 2))) return r;
        if ((r = ensure(c, t->mlp, rows * TF * 2))) return r;
        if ((r = ensure(c, t->stats, rows * 2 * sizeof(float)))) return r;
        if ((r = ensure(c, t->lnpart, rows * 2 * (TD / 64) * sizeof(float)))) return r;
        if ((r = ensure(c, t->inter, (size_t)(t->ni > 0 ? t->ni : 1) * rows * TD * 2))) return r;
        if ((r = ensure(c, t->meta, (size_t)cap * 2 * sizeof(int32_t)))) return r;
        t->ws_images = cap;
    }
    const int64_t ws_rows = (int64_t)t->ws_images * TT;
    for (int i0 = 0; i0 < n; i0 += per_pass) {
        const int m = n - i0 < per_pass ? n - i0 : per_pass;
        const int M = m * TT;
        if (i0 > 0) HIP_TRY(c, hipStreamSynchronize(s));  // the id tables of successive passes share one device buffer
        int32_t* aid_dev = (int32_t*)t->meta.p;
        int32_t* nt_dev = aid_dev + t->ws_images;
        HIP_TRY(c, hipMemcpyAsync(aid_dev, aspect_ids_host + i0, (size_t)m * 4, hipMemcpyHostToDevice, s));
        HIP_TRY(c, hipMemcpyAsync(nt_dev, num_tiles_host + i0, (size_t)m * 4, hipMemcpyHostToDevice, s));
        // one guard word per layer for the fast attention form (attention_tiles.hip): zeroed per pass
        {
            int rg;
            if ((rg = ensure(c, c->attn_guard, 64 * sizeof(int)))) return rg;
        }
        if (c->attn_mode) HIP_TRY(c, hipMemsetAsync(c->attn_guard.p, 0, 64 * sizeof(int), s));
        const int64_t npatch = (int64_t)m * TTILES * TGRID * TGRID;
        {
            Timed tm(c, s, KC_PRE);
            HIP_TRY(c, launch_tile_patchify(pixel_values + (size_t)i0 * TTILES * 3 * 560 * 560, t->patches.p, npatch, s));
        }
        GemmArgs g{};
        {
            Timed tm(c, s, KC_GEMM);
            g.A = t->patches.p; g.W = t->patch_w; g.M = (int)npatch; g.N = TD; g.K = TPDIMP; g.bias = t->zeros; g.out = t->pemb.p; g.ldo = TD;
            HIP_TRY(c, launch_gemm(EPI_BIAS, g, s, c->gemm_variant));
        }
        {
            Timed tm(c, s, KC_LN);
            HIP_TRY(c, launch_tile_assemble(t->pemb.p, t->cls, t->pre, t->pos, t->tilepos, t->lnpre_g, t->lnpre_b, aid_dev, t->x.p, M, 1e-5f, s));
        }
        auto stats_from_x = [&](int64_t row0) -> int {
            Timed tm(c, s, KC_LN);
            HIP_TRY(c, launch_ln_stats_canonical(t->x.p, row0, M, TD, t->eps, (float*)t->stats.p, s));
            return MME_OK;
        };
        auto stats_after = [&](const GemmArgs& producer) -> int {
            if (c->ln_mode != 2 || !gemm_runs_256(producer, c->gemm_variant)) return stats_from_x(0);
            const int64_t interior = (int64_t)(M / 256) * 256;
            {
                Timed tm(c, s, KC_LN);
                HIP_TRY(c, launch_ln_finish((const float*)t->lnpart.p, producer.ln_part_rows, interior, TD, t->eps, (float*)t->stats.p, s));
            }
            return interior < M ? stats_from_x(interior) : MME_OK;
        };
        const int res_epi = c->ln_mode == 2 ? EPI_BIAS_RES_STATS : EPI_BIAS_RES;
        int r;
        if ((r = stats_from_x(0))) return r;
        const int L = t->layers + t->global_layers;
        int saved = 0;
        for (int l = 0; l < L; ++l) {
            const TileLayerDev& Ld = t->layer[l];
            // an intermediate state the output concatenates, "before layer l" convention: the state ENTERING local layer l
            if (t->save_before && l < t->layers && saved < t->ni && t->inter_after[saved] == l) {
                HIP_TRY(c, hipMemcpyAsync((char*)t->inter.p + (size_t)saved * ws_rows * TD * 2, t->x.p, (size_t)M * TD * 2, hipMemcpyDeviceToDevice, s));
                ++saved;
            }
            if (l == t->layers) {  // between the local and the global stack: layernorm_post + post-tile embedding
                {
                    Timed tm(c, s, KC_LN);
                    HIP_TRY(c, launch_tile_ln_post(t->x.p, t->lnpost_g, t->lnpost_b, t->post, aid_dev, M, 1e-5f, s));
                }
                if ((r = stats_from_x(0))) return r;
            }
            {
                Timed tm(c, s, KC_GEMM);
                g = GemmArgs{};
                g.A = t->x.p; g.W = Ld.qkv_wf; g.M = M; g.N = 3 * TD; g.K = TD;
                g.bias = Ld.qkv_bf; g.colsum = Ld.qkv_cs; g.ln_stats = (const float*)t->stats.p; g.out = t->qkv.p; g.ldo = 3 * TD;
                HIP_TRY(c, launch_gemm(EPI_LN_BIAS, g, s, c->gemm_variant));
            }
            {
                Timed tm(c, s, KC_ATTN);
                HIP_TRY(c, launch_attention_tiles(t->qkv.p, t->att.p, nt_dev, m, s, c->attn_mode ? (int*)c->attn_guard.p + l : nullptr, c->attn_mode == 2));
            }
            {
                Timed tm(c, s, KC_GEMM);
                g = GemmArgs{};
                g.A = t->att.p; g.W = Ld.o_w; g.M = M; g.N = TD; g.K = TD;
                g.bias = t->zeros; g.out = t->x.p; g.res = t->x.p; g.ldo = TD;
                g.ln_part = (float*)t->lnpart.p; g.ln_part_rows = ws_rows;
                HIP_TRY(c, launch_gemm(res_epi, g, s, c->gemm_variant));
            }
            if ((r = stats_after(g))) return r;
            {
                Timed tm(c, s, KC_GEMM);
                g = GemmArgs{};
                g.A = t->x.p; g.W = Ld.fc1_wf; g.M = M; g.N = TF; g.K = TD;
                g.bias = Ld.fc1_bf; g.colsum = Ld.fc1_cs; g.ln_stats = (const float*)t->stats.p; g.out = t->mlp.p; g.ldo = TF;
                HIP_TRY(c, launch_gemm(EPI_LN_BIAS_GELU, g, s, c->gemm_variant));
            }
            // statistics are needed by the next layer's QKV GEMM, except after the last local layer (layernorm_post
            // rewrites x first) and after the very last layer
            const bool need_stats = l + 1 < L && l + 1 != t->layers;
            {
                Timed tm(c, s, KC_GEMM);
                g = GemmArgs{};
                g.A = t->mlp.p; g.W = Ld.fc2_w; g.M = M; g.N = TD; g.K = TF;
                g.bias = Ld.fc2_b; g.out = t->x.p; g.res = t->x.p; g.ldo = TD;
                g.ln_part = (float*)t->lnpart.p; g.ln_part_rows = ws_rows;
                HIP_TRY(c, launch_gemm(need_stats ? res_epi : EPI_BIAS_RES, g, s, c->gemm_variant));
            }
            if (need_stats && (r = stats_after(g))) return r;
            if (!t->save_before && l < t->layers && saved < t->ni && t->inter_after[saved] == l) {  // "after layer l" convention
                HIP_TRY(c, hipMemcpyAsync((char*)t->inter.p + (size_t)saved * ws_rows * TD * 2, t->x.p, (size_t)M * TD * 2, hipMemcpyDeviceToDevice, s));
                ++saved;
            }
        }
        Timed tm(c, s, KC_POOL);
        const int F = TD * (1 + t->ni);
        if (hidden)
            HIP_TRY(c, launch_tile_output(t->x.p, t->inter.p, t->ni, ws_rows * TD, hidden + (size_t)i0 * TTILES * TTOK * F, (int64_t)m * TTILES * TTOK, s));
        if (emb_f32 || emb_bf16)
            HIP_TRY(c, launch_tile_pool(t->x.p, t->inter.p, t->ni, ws_rows * TD, m, emb_f32 ? emb_f32 + (size_t)i0 * F : nullptr,
                                        emb_bf16 ? (void*)(emb_bf16 + (size_t)i0 * F) : nullptr, s));
    }
    return MME_OK;
}

}  // extern "C"
