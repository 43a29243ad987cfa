/* mme.h -- C ABI of the MI355X-native region embed -> compare -> page-cluster engine.
 *
 * The reference (calhounpaul/multimodal_embeddings) has no FFI: its hot path is two
 * duck-typed Python objects (an `embedder` and a vector-store `collection`) plus two
 * free functions.  This header is the boundary a maintainer binds with
 * ctypes (INTEGRATION.md shows the stub); each entry point cites the reference
 * interface it replaces.  Plain pointers and sizes only -- no torch types.
 *
 * Conventions
 *   - one mme_ctx per GPU / per process rank; not thread-safe (use one ctx per thread);
 *   - every function returns 0 on success or a negative MME_E_* code and never throws;
 *     mme_last_error(ctx) returns the message of the last failure;
 *   - "dev" pointers are device (HBM) pointers owned by the caller, "host" pointers are
 *     ordinary host memory; all launches are asynchronous on `stream` (a hipStream_t
 *     passed as void*; NULL = the default stream) unless the name ends in _sync;
 *   - bf16 = 16-bit brain float stored as uint16_t.
 */
#ifndef MME_H
#define MME_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Bumped whenever an export changes its signature, the meaning of an argument or the size of a caller-owned array.
 * 2 (round 3): mme_profile_read_sync takes the capacity of the caller's arrays (the class count is no longer part of
 *    the ABI); mme_set_ln_fusion's argument is a MODE (0 / 1 / 2, it was on / off in version 1); mme_tile_vit_weights
 *    carries the save point of the intermediate states; the experiment switches MME_GEMM_DEBUG / MME_ATTN_DEBUG exist
 *    only in a -DMME_DIAG build; new exports mme_is_diag_build, mme_set_attention_mode, mme_attention_redone,
 *    mme_set_tile_order, mme_set_forward_pruning.  A binder checks `mme_abi_version() == MME_ABI_VERSION` right after dlopen. */
#define MME_ABI_VERSION 2

enum {
    MME_OK = 0,
    MME_E_ARG = -1,     /* bad argument (null pointer, size, alignment) */
    MME_E_STATE = -2,   /* call order (e.g. forward before load) */
    MME_E_HIP = -3,     /* HIP runtime / launch failure */
    MME_E_NOMEM = -4,
    MME_E_COMM = -5     /* RCCL missing or a collective failed */
};

typedef struct mme_ctx mme_ctx;

/* ---- lifetime ---------------------------------------------------------------------
 * Replaces MmE5MllamaEmbedder.__init__'s per-device replica set-up
 * (deprecated_package/embedder.py:42-84): one context per visible GPU. */
int mme_abi_version(void);
/* 1 when the library was built with -DMME_DIAG (libmme_diag.so): only then are the experiment switches of DESIGN.md
 * 4.5 (MME_GEMM_DEBUG, MME_GEMM_GN / RB / GRID / MIN256, MME_ATTN_BUFS / PIPE / DEBUG, MME_K1_VWIN / HBAND) read from
 * the environment.  The production library ignores them. */
int mme_is_diag_build(void);
int mme_create(int device, mme_ctx** out);
void mme_destroy(mme_ctx* ctx);
const char* mme_last_error(const mme_ctx* ctx); /* ctx may be NULL: creation errors */

/* ---- encoder weights --------------------------------------------------------------
 * Replaces `MllamaForConditionalGeneration.from_pretrained(...)` (embedder.py:75-80) for
 * the re-scoped ViT-B/16 encoder.  Host f32 tensors in Hugging Face ViT layout
 * (transformers models/vit/modeling_vit.py): Linear weights are [out, in]; the patch
 * projection is [hidden, 3*patch*patch] in (c, ky, kx) order.  Values are rounded to
 * bf16 on upload (the reference runs the encoder in bf16, embedder.py:78). */
typedef struct {
    const float *ln1_g, *ln1_b;
    const float *q_w, *q_b, *k_w, *k_b, *v_w, *v_b, *o_w, *o_b;
    const float *ln2_g, *ln2_b;
    const float *fc1_w, *fc1_b, *fc2_w, *fc2_b;
} mme_vit_layer;

typedef struct {
    int32_t image_size;  /* 224 */
    int32_t patch_size;  /* 16  */
    int32_t hidden;      /* 768 */
    int32_t layers;      /* 12  */
    int32_t heads;       /* 12  */
    int32_t mlp;         /* 3072 */
    float ln_eps;        /* 1e-12 */
    const float* cls_token; /* [hidden] */
    const float* pos_emb;   /* [1 + (image/patch)^2, hidden] */
    const float* patch_w;   /* [hidden, 3*patch*patch] */
    const float* patch_b;   /* [hidden] */
    const float* lnf_g;     /* final LayerNorm */
    const float* lnf_b;
    const mme_vit_layer* layer; /* [layers] */
} mme_vit_weights;

int mme_load_vit(mme_ctx* ctx, const mme_vit_weights* w);

/* Pixel normalisation constants of the image processor (per channel; default CLIP).
 * Replaces the `image_mean` / `image_std` of the checkpoint's preprocessor_config. */
int mme_set_normalisation(mme_ctx* ctx, const float mean[3], const float std[3]);

/* Rows of the internal activation workspace = crops per encoder pass (default 4096: one pass
 * for the headline batch; 11.6 GB of workspace; larger passes lose less to tile quantisation). */
int mme_set_chunk(mme_ctx* ctx, int crops_per_pass);

/* Tuning / test knob: which MFMA GEMM tiling serves K2/K4/K6/K7/K9.  0 = by shape (default),
 * 1 = 128x128 tiles, 3 = 256x256 ping-pong kernel with a 3-deep activation ring (2 is accepted and
 * means 3), 4 / 6 / 5 = variant 3 with 4 / 6 / 8 of a lane's 16 output stores deferred into the next
 * tile's first K-tile (4 is the default for large problems).  Results are bit-identical across
 * variants (same MFMA instruction, same K order per output element). */
int mme_set_gemm_variant(mme_ctx* ctx, int variant);

/* LayerNorm folding: LN1 / LN2 are folded into the QKV / fc1 GEMMs
 * (W' = W*gamma, out = rstd*(W'x - mean*colsum) + b'), so no normalised copy is written.
 *   2 (default) the per-row statistics come from partial sums the GEMM that wrote the residual
 *     stream left behind (96 bytes per row to finish; the stream is not read again);
 *   1 one statistics pass over the residual stream per LayerNorm, same canonical summation order
 *     (bit-identical embeddings to mode 2);
 *   0 separate LayerNorm kernel (A/B, tests). */
int mme_set_ln_fusion(mme_ctx* ctx, int mode);

/* K5 (attention of the ViT-B/16 forward, transformers modeling_vit.py:164-189).
 *   1 (default) fast form: the exponentials of a query row are taken against the maximum over its first 32 keys instead
 *     of its row maximum -- softmax is invariant to that choice, only the range differs -- which lets the scores leave
 *     the matrix pipe ready for exp2.  A row whose sum leaves [1, 2^100) raises a per-launch guard word and the
 *     launch is redone by the exact kernel (the decision is taken on the device; the call stays asynchronous), so
 *     every finite input gets the exact algorithm's result;
 *   0 exact form only (row maximum first), as the reference computes it.
 *   2 the fast form with the guard forced for every row: every launch is redone by the exact kernel (a test of the
 *     re-run path: outputs are bit-identical to mode 0).
 * Outputs of modes 0 and 1 agree to rounding (different rounding points of the probabilities), not bit for bit.
 * The query projection carries dh^-0.5 log2(e) in every mode (folded into W_q / b_q by mme_load_vit).
 * Granularity and cost of the guard: ONE word per layer launch, so a single query row out of range re-runs that layer's
 * attention for EVERY crop of the pass (results stay correct; the launch then costs fast + exact).  The range is wide -- raw
 * scores q.k / sqrt(d) 550 apart within a row -- and seeded weights trip it only with W_q, W_k scaled x7 and more
 * (profiles/round3_fuzz_attention.txt); the redo rate on a TRAINED checkpoint is unmeasured (none is available offline):
 * mme_attention_redone says which layers of the last pass were redone, bench.py prints their count next to the headline,
 * and mode 0 is the setting for a checkpoint that trips the guard routinely.
 * The same switch governs the tile-ViT encoder's attention (mme_tile_vit_forward; attention_tiles.hip), whose fast form
 * differs: the reference point of a row starts as the maximum over its first 32 keys and is RE-CENTRED from the row sum after
 * every 128-key tile (a sum past 2^60 moves the reference by the sum's exponent: exact powers of two), so the guard fires only
 * when a score jumps ~67 log2 units (46 nats of q.k / sqrt(d)) above everything the row met before within one tile -- the sum
 * is then inf / NaN and that layer's launch is redone by the exact kernel (one guard word per layer, 40 for the full tower). */
int mme_set_attention_mode(mme_ctx* ctx, int mode);
/* Order in which the kernels of an encoder pass walk the rows of the activations.  1 (default) zig-zag: consecutive kernels
 * walk in opposite directions, so a consumer starts on the rows its producer wrote last -- what is still in the 256 MiB
 * Infinity Cache of a 1.2-5 GB activation; 0 every kernel upwards; 2 only the attention downwards.  Tile order only: results
 * are bit-identical in every mode.  Worth 0.3-0.6 ms per step on boxes whose HBM streams at 3.9 TB/s, nothing on the others. */
int mme_set_tile_order(mme_ctx* ctx, int mode);
/* Last-layer pruning (default OFF; no reference counterpart -- the reference computes the whole last hidden state and
 * `last_pooling` then reads ONE token row of it, embedder.py:17-34).  With it on, the rows nothing reads are not computed:
 * in the last layer only the query block that holds the pooled token is attended, and its o_proj / LayerNorm / MLP run on
 * the n gathered rows instead of n x 197 (6.2 % of the forward's FLOP).  Same kernels, same per-row arithmetic: the
 * embeddings are bit-identical to the full pass.  Off by default so that the headline benchmark times the WHOLE forward
 * (bench.py reports the pruned rate separately); applies to the LayerNorm-folded modes (mme_set_ln_fusion 1 / 2). */
int mme_set_forward_pruning(mme_ctx* ctx, int on);
/* Diagnostic (synchronises the device): flags[l] != 0 when the attention launch of layer l of the LAST encoder pass
 * raised its guard and was redone by the exact kernel. */
int mme_attention_redone(mme_ctx* ctx, int32_t flags[12]);

/* ---- K0: cut the bounding boxes of one decoded page on the device ----------
 * Replaces DocLayoutDetector.get_region_image (doclayout_detector.py:165-194), which re-opens
 * and re-decodes the whole page PNG for every region: the page is uploaded once and every box
 * is gathered into the packed crop buffer that mme_preprocess / mme_embed read.
 *   page_dev    uint8 RGB HWC pixels of the page, [H, W, 3]
 *   boxes_host  int32[n,4] (x0, y0, x1, y1) AFTER the reference's int() truncation
 *               (doclayout_detector.py:179); crop i is (y1-y0) x (x1-x0) pixels; parts of a box
 *               outside the page read as 0, as PIL's Image.crop fills them
 *   pix_dev     destination buffer; offs_host int64[n] byte offset of crop i in it (same
 *               packing rule as mme_preprocess: 16-byte aligned crops, 16 bytes of slack) */
int mme_crop_boxes(mme_ctx* ctx, const uint8_t* page_dev, int H, int W, const int32_t* boxes_host, int n, uint8_t* pix_dev,
                   const int64_t* offs_host, void* stream);

/* ---- K13: merge the detector's grid passes -- class-aware non-maximum suppression ----------
 * Replaces apply_non_max_suppression / calculate_iou (3_combine_grids.py:44-137), the O(n^2) list.index / list.pop loop
 * that turns the boxes of all grid passes of a page into the page's region list: keep the highest-scoring box left (the
 * first of equal scores), drop every remaining box of the same class whose IoU with it exceeds iou_threshold.  Many pages
 * per call, one workgroup each; float64 in the reference's operation order, so the kept set and its order are identical.
 * All pointers are HOST memory (the boxes come from JSON and the result goes back into JSON); the call returns when
 * the results are there.
 *   boxes f64[n,4] (x0,y0,x1,y1), scores f64[n], classes int32[n]; page p owns rows [page_offs[p], page_offs[p+1]),
 *   page_offs int32[pages+1] with page_offs[0] = 0, at most 32768 boxes per page
 *   keep int32[n]: for page p, keep[page_offs[p] + k] = page-local index of the k-th kept box in the reference's output
 *   order, -1 beyond keep_count[p]; keep_count int32[pages] */
int mme_nms_boxes(mme_ctx* ctx, const double* boxes, const double* scores, const int32_t* classes, const int32_t* page_offs, int pages,
                  double iou_threshold, int32_t* keep, int32_t* keep_count, void* stream);

/* ---- K1: crop -> resize -> pad -> normalise -> patchify ------------------------------
 * Replaces, per crop, `processor(images=[image])` (embedder.py:117-121; transformers
 * image_processing_pil_mllama.py:483-541 with tile 224, one tile): aspect-preserving
 * Pillow-BILINEAR fit (bit-exact incl. the per-pass uint8 rounding), zero pad right/
 * bottom BEFORE normalisation, x/255, (x-mean)/std, im2col to [196, 768] in (c,ky,kx).
 *   pix_dev     uint8 RGB HWC pixels of all crops, concatenated; the allocation must
 *               extend at least 16 bytes past the last pixel (rows are read in words)
 *   offs_host   int64[n]   byte offset of crop i inside pix_dev
 *   hw_host     int32[n,2] (height, width) of crop i  (int()-truncated bbox size,
 *                          doclayout_detector.py:179)
 *   patches_dev bf16[n*196, 768] */
int mme_preprocess(mme_ctx* ctx, const uint8_t* pix_dev, const int64_t* offs_host, const int32_t* hw_host,
                   int n, uint16_t* patches_dev, void* stream);

/* Mllama-faithful multi-tile preprocessing: what `processor(images=[image])`
 * (embedder.py:117-121) computes with the checkpoint's geometry -- transformers
 * image_processing_pil_mllama.py:483-541: choose the tile canvas among all grids of <= max_tiles
 * tiles (:299-355), aspect-preserving Pillow-BILINEAR fit into it (:246-295, :431-481), zero pad to
 * the canvas, x/255, (x-mean)/std (the values of mme_set_normalisation), split into tiles row-major
 * (:39-49), zero-pad the tile axis (:84-133).  Bit-exact f32.
 *   pix_dev / offs_host / hw_host   as mme_preprocess
 *   tile, max_tiles                 560 and 4 for mmE5-mllama; tile % 8 == 0
 *   out_dev            float[n, max_tiles, 3, tile, tile]  (`pixel_values`)
 *   aspect_ids_host    int32[n] or NULL: `aspect_ratio_ids` (1-based index into the supported grids, :136-164)
 *   num_tiles_host     int32[n] or NULL: tiles used; `aspect_ratio_mask` = 1 for the first num_tiles slots (:52-81)
 * Synchronises the stream once (crop tables are staged from host temporaries). */
int mme_preprocess_tiles(mme_ctx* ctx, const uint8_t* pix_dev, const int64_t* offs_host, const int32_t* hw_host, int n, int tile,
                         int max_tiles, float* out_dev, int32_t* aspect_ids_host, int32_t* num_tiles_host, void* stream);

/* ---- K2-K8: ViT forward + pool + L2 normalise -------------------------------------------
 * Replaces `model(**inputs, output_hidden_states=True)` + `last_pooling`
 * (embedder.py:124-129, :17-34).  pool_token: 0 = [CLS] (default), 196 = last token
 * (the reference's "last attended token" with an all-ones mask), any 0..196.
 *   emb_f32_dev  float[n, hidden]  L2-normalised (may be NULL)
 *   emb_bf16_dev bf16 [n, hidden]  same vectors rounded to bf16 (may be NULL) */
int mme_vit_forward(mme_ctx* ctx, const uint16_t* patches_dev, int n, int pool_token,
                    float* emb_f32_dev, uint16_t* emb_bf16_dev, void* stream);

/* K1 + K2-K8 in one call, chunked through the internal workspace.  This is the device
 * side of `get_image_embeddings` (embedder.py:141-226) for decoded crops. */
int mme_embed(mme_ctx* ctx, const uint8_t* pix_dev, const int64_t* offs_host, const int32_t* hw_host,
              int n, int pool_token, float* emb_f32_dev, uint16_t* emb_bf16_dev, void* stream);

/* f32 vectors (the reference moves embeddings as Python float lists, embedder.py:132) ->
 * L2-normalised bf16 rows, same normalisation as last_pooling (F.normalize, eps 1e-12).
 * x_dev float[rows,d] (d % 4 == 0, 16-byte aligned rows), y_dev bf16[rows,d]. */
int mme_normalise_rows(mme_ctx* ctx, const float* x_dev, int64_t rows, int d, uint16_t* y_dev, void* stream);

/* ---- K9: all-pairs cosine ---------------------------------------------------------------
 * The exact object every `collection.query` of the reference samples from
 * (weighted_region_clustering.py:79-84, region_compare.py:165-170,
 * cross_compare.py:119-123): sim[i,j] = <a_i, b_j> over L2-normalised bf16 rows,
 * f32 accumulate.  a [m,d], b [n,d] row-major, d % 64 == 0; sim f32 row-major with
 * leading dimension ld_sim >= n. */
int mme_cosine(mme_ctx* ctx, const uint16_t* a_dev, int m, const uint16_t* b_dev, int n, int d,
               float* sim_dev, int64_t ld_sim, void* stream);

/* The same block with S rounded to bf16 (round-to-nearest-even of the f32 accumulator: |error| <= 2^-9 relative, i.e.
 * <= 0.002 on a cosine): halves the bytes the compare stage writes and the next stage reads -- the [8192 x 65536] block of
 * one rank of config C4 is 1.07 GB instead of 2.15 GB.  n % 4 == 0 and ld_sim % 8 == 0 (16-byte stores).  Ranking
 * consumers that must match the f32 order (K10 / K12) keep using the f32 values; this is the output option for callers
 * that store or threshold similarities. */
int mme_cosine_bf16(mme_ctx* ctx, const uint16_t* a_dev, int m, const uint16_t* b_dev, int n, int d, uint16_t* sim_dev, int64_t ld_sim,
                    void* stream);

/* ---- K10: segmented top-k + area-weighted page reduction ---------------------------------
 * Replaces the page-pair loop of compute_image_similarity_matrix
 * (weighted_region_clustering.py:162-252).  Regions are grouped by page:
 * rows page_offs[p] .. page_offs[p+1]-1 of emb belong to page p, in collection order.
 *   emb_dev        bf16[N, d] L2-normalised region vectors
 *   area_pct_dev   double[N]  area_percentage (0-100) as stored (region_processor.py:89-93)
 *   valid_dev      uint8[N]   1 = area>0 and type in REGION_TYPES_TO_PROCESS (wrc:136)
 *   page_offs_host int32[P+1]
 *   skip_dev       uint8[P,P] 1 = pair skipped (same 20-char prefix, wrc:179-186); may be NULL
 *   max_query      10 (wrc:199), top_k 10 (wrc:210), max_dist 0.9 (wrc:223)
 *   metric         0: d = 1-cos, 1: d = 2-2cos
 *   normalise      1: divide off-diagonal by its max, diagonal = 1 (wrc:246-252)
 *   S_dev          double[P,P] */
int mme_page_similarity(mme_ctx* ctx, const uint16_t* emb_dev, int64_t N, int d, const double* area_pct_dev,
                        const uint8_t* valid_dev, const int32_t* page_offs_host, int P, const uint8_t* skip_dev,
                        int max_query, int top_k, double max_dist, int metric, int normalise, double* S_dev,
                        void* stream);

/* Multi-GPU form: compute only the page pairs whose rank in the row-major upper triangle
 * (i < j) lies in [pair_lo, pair_hi), un-normalised, every other entry of S_dev = 0.  Ranks split
 * 0..P(P-1)/2 evenly, add their S (disjoint entries: the sum is exact) and normalise once. */
int mme_page_similarity_pairs(mme_ctx* ctx, const uint16_t* emb_dev, int64_t N, int d, const double* area_pct_dev,
                              const uint8_t* valid_dev, const int32_t* page_offs_host, int P, const uint8_t* skip_dev,
                              int max_query, int top_k, double max_dist, int metric, int64_t pair_lo, int64_t pair_hi,
                              double* S_dev, void* stream);

/* ---- K11: page clustering -------------------------------------------------------------------
 * Replaces cluster_images' arithmetic (weighted_region_clustering.py:476-543): average
 * linkage + silhouette-chosen k.  S_dev double[P,P] page similarities with unit diagonal
 * (2 <= P <= 4096).
 *   n_clusters  0 = choose k in 2..min(10,P) (or ..min(3,P) when fewer than 10 entries of S
 *               exceed 0.01 off the diagonal, wrc:482-490) by silhouette, strict-> argmax
 *               from -1 (wrc:492,517); >0 = cut at that k (wrc:531-543)
 *   mode        0 = what scikit-learn >= 1.4 executes through the reference's TypeError
 *               fallback (wrc:504-509): euclidean metric over the ROWS of D = 1-S -- the
 *               path the bundled golden labels pin; 1 = D as a precomputed distance matrix
 *   labels_dev  int32[P]  sklearn label numbering (_hc_cut heap order)
 *   k_dev       int32[1]  chosen number of clusters
 *   scores_dev  double[16]: scores[k] = silhouette at k (NaN where not evaluated) */
int mme_cluster_pages(mme_ctx* ctx, const double* S_dev, int P, int n_clusters, int mode, int32_t* labels_dev,
                      int32_t* k_dev, double* scores_dev, void* stream);

/* K12 ranked neighbour lists.  Replaces the query + filter loops of
 * create_region_cross_comparison (region_compare.py:160-353) and create_cross_comparison
 * (cross_compare.py:109-235): for each query row r in [row0, row0 + nrows) of the L2-normalised
 * bf16 rows emb_dev[N, d], take the `fetch` nearest rows by cosine -- r itself included, as the
 * store's query returns it; order = stable ascending distance, i.e. similarity descending, index
 * ascending on ties -- drop r unless keep_self (region_compare.py:244), drop rows whose group id
 * equals r's (same parent page, :260; group_dev may be NULL), drop similarities outside
 * [min_sim, max_sim] (:269), keep the first top_n (:352).
 *   fetch, top_n  1..128 (the reference uses fetch = min(3 top_n, 100), top_n = 10)
 *   idx_dev   int32[nrows, top_n] row indices, -1 padded;  sim_dev  float[nrows, top_n] cosine
 * Small N: the [rows, N] cosine block is produced chunk-wise by the MFMA GEMM into an internal workspace
 * (<= 2 GiB) and consumed by a one-wave-per-row streaming top-k.  N >= 16384: fused, the block is never written
 * (mme_set_neighbour_mode).  S is never materialised either way.
 * Multi-GPU: each rank passes its own [row0, row0 + nrows) against the all-gathered emb. */
int mme_neighbours(mme_ctx* ctx, const uint16_t* emb_dev, int N, int d, const int32_t* group_dev, int row0, int nrows,
                   int fetch, int top_n, int keep_self, float min_sim, float max_sim, int32_t* idx_dev, float* sim_dev,
                   void* stream);

/* K12 execution form: 0 = by size (default), 1 = the [rows, N] cosine block goes through the workspace in chunks,
 * 2 = fused (N >= 16384): a sampled per-row threshold, then the cosine GEMM appends only the values above it to
 * per-row candidate lists -- the block is never written; lists that overflow re-run their chunk in form 1 on the
 * device's own decision.  Results are identical. */
int mme_set_neighbour_mode(mme_ctx* ctx, int mode);

/* Diagnostic: time one MFMA GEMM shape on random bf16 data (allocates its own operands;
 * synchronous).  epilogue 0 bias, 1 bias+GELU, 2 bias+residual, 3 patch-embed, 4 f32 out;
 * variant as mme_set_gemm_variant. */
int mme_gemm_bench(mme_ctx* ctx, int M, int N, int K, int epilogue, int variant, int iters, double* avg_ms);

/* Diagnostic: run the stamped build of the 256x256 3-deep-ring GEMM (bias epilogue) once on random data.
 * stamps_host uint64[256 workgroups][2 waves (0 and 4)][16]: s_memtime cycles summed over the K-tiles the
 * wave processed -- [0..7] the eight barrier-to-barrier intervals of a K-tile, [8] time in the counted wait,
 * [9] everything between two tiles' K loops, [10] K-tiles processed, [11] of [9]: group sync + next tile's
 * prologue issue, [12] of [9]: epilogue body (loads, math, store issue); the rest of [9] is the wait for the
 * next tile's first K-tile; [13] s_memtime cycles and [14] s_memrealtime ticks (100 MHz) of the whole kernel:
 * [13] / [14] x 100 MHz is the clock the chip held (the call runs ~0.5 s of the product kernel first). */
int mme_gemm_stamps(mme_ctx* ctx, int M, int N, int K, uint64_t* stamps_host);

/* Diagnostic: time the attention kernel (K5) on B crops of random activations (avg_ms over iters launches), then run
 * its stamped build once.  stamps_host uint64[B workgroups][8 waves][8]: s_memtime cycles summed over the 12 head
 * iterations of the wave -- [0] wait for its own requests, [1] workgroup barrier, [2] issue of the next head's
 * requests (K/V LDS-DMA on wave 7, Q prefetch on the others), [3] S^T = K.Q^T, [4] softmax, [5] O^T = V^T.P^T,
 * [6] hand-over + output stores; [7] heads processed.  Wave 7 (staging only) carries in [5] / [6] the s_memtime cycles and
 * s_memrealtime ticks (100 MHz) of the whole workgroup: [5] / [6] x 100 MHz = the clock the chip held. */
int mme_attention_stamps(mme_ctx* ctx, int B, int iters, double* avg_ms, uint64_t* stamps_host);

/* ---- tile-ViT encoder option: the reference encoder's own vision-tower geometry -------------------
 * Replaces the vision side of `MllamaForConditionalGeneration.from_pretrained(...)` / `model(**inputs)`
 * (deprecated_package/embedder.py:75-79,117-126): transformers `MllamaVisionModel` at the checkpoint's configuration
 * (config.py:58; configuration_mllama.py:61-82 at image_size 560) -- <= 4 tiles of 560 x 560, patch 14, 1 + 1600 tokens
 * per tile padded to 1608, ONE sequence of 6432 tokens per image, 1280-d, 16 heads of 80, MLP 5120, `layers` local +
 * `global_layers` tanh-gated layers (32 + 8), the states after `intermediate[]` local layers (3, 7, 15, 23, 30)
 * concatenated behind the final state -> 1280 * (1 + n_intermediate) = 7680 features per token.  Geometry is fixed
 * at compile time; the layer counts are free (tests run shallow stacks against the CPU oracle).
 * Host f32 tensors in the Hugging Face state-dict layout: Linear weights [out, in]; patch_w [1280, 3*14*14] in
 * (c, ky, kx) order; pos_emb [1601, 1280]; tile_pos_emb [9, 4*1601*1280]; pre_emb / post_emb [9, 4*1280].
 * Values are rounded to bf16 on upload; the tanh gates are folded into the tables / weights they scale. */
enum { MME_TILE_SAVE_AFTER_LAYER = 0, MME_TILE_SAVE_BEFORE_LAYER = 1 };

typedef struct {
    const float *ln1_g, *ln1_b;           /* input_layernorm */
    const float *q_w, *k_w, *v_w, *o_w;   /* no biases */
    const float *ln2_g, *ln2_b;           /* post_attention_layernorm */
    const float *fc1_w, *fc1_b, *fc2_w, *fc2_b;
    float gate_attn, gate_ffn;            /* raw parameters; tanh is applied by the library */
    int32_t gated;                        /* 1 for the global stack */
} mme_tile_layer;

typedef struct {
    int32_t image_size;  /* 560 */
    int32_t patch_size;  /* 14 */
    int32_t hidden;      /* 1280 */
    int32_t heads;       /* 16 */
    int32_t mlp;         /* 5120 */
    int32_t max_tiles;   /* 4 */
    int32_t aspect_ratios; /* 9 = max_aspect_ratio_id + 1 */
    int32_t layers, global_layers;
    int32_t n_intermediate;
    int32_t intermediate[8];
    /* Which state `intermediate[k] = i` names.  MME_TILE_SAVE_AFTER_LAYER: the OUTPUT of local layer i -- what
     * transformers 5.15's MllamaVisionEncoder collects (`encoder_states` is appended after each layer; the version the
     * oracle and tests/golden/tile_vit_cases.npz are pinned to).  MME_TILE_SAVE_BEFORE_LAYER: the state ENTERING local
     * layer i (= the output of layer i - 1; i = 0: the embeddings after layernorm_pre) -- the convention of encoders that
     * record the state before running a layer (reported for the first Mllama releases around transformers 4.45 and the
     * original model code; NOT verifiable offline, "parity unpinned": tested against the oracle's own restatement only).
     * A binder picks the convention of the transformers version its checkpoint's features were produced with. */
    int32_t intermediate_save_point;
    float norm_eps;      /* 1e-5 (the encoder layers; layernorm_pre / _post use torch's default 1e-5 too) */
    float pos_gate, pre_gate, post_gate;
    const float* class_embedding;
    const float* patch_w;
    const float* pos_emb;
    const float* tile_pos_emb;
    const float* pre_emb;
    const float* post_emb;
    const float *ln_pre_g, *ln_pre_b, *ln_post_g, *ln_post_b;
    const mme_tile_layer* layer; /* [layers + global_layers], local stack first */
} mme_tile_vit_weights;

int mme_load_tile_vit(mme_ctx* ctx, const mme_tile_vit_weights* w);

/* pixel_values_dev f32 [n, 4, 3, 560, 560] (what mme_preprocess_tiles writes), aspect_ids_host / num_tiles_host int32[n]
 * (its other two outputs).  Any of the three outputs may be NULL:
 *   hidden_dev    f32 [n, 4, 1601, F]   `last_hidden_state` of the vision model (padding tiles included, as the model returns them)
 *   emb_f32_dev   f32 [n, F], emb_bf16_dev bf16 [n, F]   the class token of tile 0, L2-normalised: the crop's vector for
 *                 the compare stage (last_pooling's rule -- one token row, F.normalize -- embedder.py:17-34).
 * F = 1280 * (1 + n_intermediate).  Images are processed mme_set_chunk (<= 64) at a time. */
int mme_tile_vit_forward(mme_ctx* ctx, const float* pixel_values_dev, const int32_t* aspect_ids_host, const int32_t* num_tiles_host, int n,
                         float* hidden_dev, float* emb_f32_dev, uint16_t* emb_bf16_dev, void* stream);

/* ---- multi-GPU: the ONE exchange step of the path ------------------------------------------
 * Replaces the hand-back of per-device results through Python lists by the reference's thread pool
 * (deprecated_package/embedder.py:208-224): every rank embeds its contiguous block of the corpus and the
 * [rows, d] bf16 shards are all-gathered over RCCL (xGMI) so that each rank can compute its row block of the
 * cosine matrix / its page pairs / its neighbour lists against all N rows.
 *   mme_comm_unique_id  rank 0 makes the 128-byte rendezvous id and hands it to the other ranks by any
 *                       means (a file, MPI, a socket);
 *   mme_comm_init       every rank, collectively: communicator for this context's GPU;
 *   mme_allgather       all[r * rows .. (r+1) * rows) = rank r's shard, asynchronous on `stream`;
 *                       equal shards (pad the last rank's block: rows are independent);
 *   mme_comm_destroy.
 * `comm` is an ncclComm_t: a communicator the caller created itself with RCCL works as well.  RCCL is resolved
 * at run time (the copy PyTorch loaded, else librccl.so.1; MME_RCCL_LIB overrides) -- libmme.so does not link it,
 * and a missing RCCL only fails these four calls (MME_E_COMM). */
#define MME_COMM_ID_BYTES 128
int mme_comm_unique_id(mme_ctx* ctx, uint8_t id_host[MME_COMM_ID_BYTES]);
int mme_comm_init(mme_ctx* ctx, const uint8_t id_host[MME_COMM_ID_BYTES], int rank, int world, void** comm_out);
int mme_comm_destroy(mme_ctx* ctx, void* comm);
int mme_allgather(mme_ctx* ctx, void* comm, const uint16_t* shard_dev, int64_t rows, int d, uint16_t* all_dev, void* stream);

/* ---- timing of the kernels by class (HIP events on the launch stream) ----------------------
 * class ids: 0 preprocess, 1 gemm, 2 layernorm, 3 attention, 4 pool, 5 cosine, 6 page_reduce, 7 cluster,
 * 8 neighbours, 9 all-gather */
#define MME_NUM_KERNEL_CLASSES 10
int mme_profile_enable(mme_ctx* ctx, int on);
int mme_profile_reset(mme_ctx* ctx);
/* synchronises the recorded events; ms[c] = total ms, launches[c] = launch count per class, for the first
 * min(count, MME_NUM_KERNEL_CLASSES) classes: `count` is the capacity of BOTH caller arrays, so a binder built against
 * a header with fewer classes is never overrun.  Returns the number of classes the library knows (>= 0) or MME_E_*. */
int mme_profile_read_sync(mme_ctx* ctx, int count, double* ms, int64_t* launches);

#ifdef __cplusplus
}
#endif
#endif /* MME_H */
