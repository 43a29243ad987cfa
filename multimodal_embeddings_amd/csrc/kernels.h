// Launcher declarations shared between the kernel translation units and capi.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint16_t bf16_bits;  // host-visible alias; device code uses __bf16

enum GemmEpilogue {
    EPI_BIAS = 0,       // out = bf16(acc + bias[n])                                  (QKV)
    EPI_BIAS_GELU = 1,  // out = bf16(gelu_erf(acc + bias[n]))                        (fc1)
    EPI_BIAS_RES = 2,   // out = bf16(acc + bias[n] + res[m,n])   (res may alias out) (o_proj, fc2)
    EPI_PATCH = 3,      // row m=(b,p) -> out row b*197+1+p, + bias[n] + pos[1+p,n]   (patch embed)
    EPI_F32 = 4,        // outf[m,n] = acc                                            (cosine)
    // LayerNorm folded into the GEMM that consumes it (A = the raw residual stream x, W' = W*gamma):
    //   out = bf16(rstd[m] * (acc - mean[m] * colsum[n]) + bias'[n])             (QKV)
    EPI_LN_BIAS = 5,
    EPI_LN_BIAS_GELU = 6, // same, then erf-GELU                                      (fc1)
    EPI_TOPK = 7,         // no output matrix: every acc[m,n] >= thr[m] is appended to row m's candidate list (K12)
    // EPI_BIAS_RES that also leaves, per output row and 64-column slice, the LayerNorm partial sums (sum x, sum x^2 of
    // the ROUNDED outputs) in ln_part: the next LayerNorm's statistics then need no pass over the residual stream
    EPI_BIAS_RES_STATS = 8
};

struct GemmArgs {
    const void* A;   // bf16 [M, K] row-major
    const void* W;   // bf16 [N, K] row-major ("TN": both operands K-contiguous)
    int M, N, K;     // valid extents; K % 64 == 0
    const float* bias;
    void* out;       // bf16
    int64_t ldo;
    const void* res; // bf16, same layout as out
    const float* pos;
    float* outf;
    int64_t ldf;
    const float* ln_stats;  // [M,2] (mean, rstd) per row, for EPI_LN_*
    float* ln_part;         // EPI_BIAS_RES_STATS: [2][N/64][ln_part_rows] f32 partial (sum, sum of squares) planes
    int64_t ln_part_rows;
    const float* colsum;    // [N] sum_k W'[n,k], for EPI_LN_*
    // EPI_TOPK: per-row threshold thr[m * thr_stride]; candidates (value, column) of row m go to
    // cand_val / cand_idx [m * cand_cap + k], k = atomic slot from cand_count[m]; a full list raises *overflow
    const float* thr;
    int thr_stride;
    int* cand_count;
    float* cand_val;
    int* cand_idx;
    int cand_cap;
    int* overflow;
    const int* run_if;      // optional: the whole launch is a no-op unless *run_if != 0 (device-side fallback switch)
    int reverse_m;          // 256 x 256 kernel: walk the row panels from the LAST one down (same results; see forward_chunk, zig-zag)
};

// raises hipFuncAttributeMaxDynamicSharedMemorySize of `kernel` on the CURRENT device to at least `bytes`
// (once per (device, kernel), thread-safe: launch_state.hip)
hipError_t ensure_dynamic_lds(const void* kernel, int bytes);

hipError_t launch_gemm(int epilogue, const GemmArgs& g, hipStream_t s, int variant = 0);
// diagnostic: stamped build of the 3-deep-ring 256x256 kernel (bias epilogue), stamps uint64[256 * 2 * 16]
hipError_t launch_gemm256r_stamped(const GemmArgs& g, unsigned long long* stamps, hipStream_t s);

// LayerNorm over rows of 768 bf16 (f32 statistics), bf16 out.
hipError_t launch_layernorm(const void* x, const float* gamma, const float* beta, void* y, int64_t rows, float eps, hipStream_t s);
// per-row LayerNorm statistics of bf16 rows of 768: stats[row] = (mean, rstd)
hipError_t launch_ln_stats(const void* x, int64_t rows, float eps, float* stats, hipStream_t s);
// the same statistics in the CANONICAL summation order shared with the EPI_BIAS_RES_STATS epilogue, for rows
// [row0, row1) of bf16 rows of `d` (d % 64 == 0, d <= 2048): one pass over x
// rows row0, row0 + stride, ... < row1
hipError_t launch_ln_stats_canonical(const void* x, int64_t row0, int64_t row1, int d, float eps, float* stats, hipStream_t s, int64_t stride = 1);
// finishes rows [0, rows) from the partial planes an EPI_BIAS_RES_STATS GEMM left: part [2][d/64][part_rows]
hipError_t launch_ln_finish(const float* part, int64_t part_rows, int64_t rows, int d, float eps, float* stats, hipStream_t s);
// true when launch_gemm(variant) runs the 256 x 256 kernel (whose fast-path epilogue writes the partial planes)
bool gemm_runs_256(const GemmArgs& g, int variant);
// x[b*197 + 0, :] = bf16(cls + pos[0])
hipError_t launch_cls_rows(void* x, const float* cls, const float* pos, int B, hipStream_t s);
// final LayerNorm on row b*197+tok, L2 normalise, write f32 and/or bf16
hipError_t launch_pool(const void* x, const float* gamma, const float* beta, int B, int tok, float eps,
                       float* emb_f32, void* emb_bf16, hipStream_t s);
// f32 [rows,d] -> L2-normalised bf16 [rows,d]
hipError_t launch_normalise_rows(const float* x, int64_t rows, int d, void* y, hipStream_t s);
// fused multi-head attention, T=197, dh=64, 12 heads; qkv [B*197, 2304] -> out [B*197, 768]
// guard: device int, zero before the launch; non-null selects the FAST kernel + the conditional exact re-run (attention.hip)
// force_redo: the fast kernel raises the guard for every row (test of the re-run path)
// only_block >= 0: compute and store that query block of 32 only (0..6)
// reverse: walk the crops from the last one down (same results; zig-zag order of consecutive kernels, forward_chunk)
hipError_t launch_attention(const void* qkv, void* out, int B, hipStream_t s, int* guard = nullptr, bool force_redo = false, int only_block = -1,
                            bool reverse = false);
// diagnostic: stamped build, stamps uint64[B][8][8]
hipError_t launch_attention_stamped(const void* qkv, void* out, int B, unsigned long long* stamps, hipStream_t s);

struct CropDesc {  // one per crop, built on the host by capi
    int64_t src_off;   // byte offset into pix
    int64_t tmp_off;   // byte offset (16-aligned) into the horizontal-pass scratch (if the width changes)
    int64_t tab_off;   // byte offset (16-aligned) of this crop's resampling tables in the table buffer (K1Layout)
    int32_t h, w;      // source size
    int32_t new_h, new_w;
};
struct HWork {  // one block of the horizontal pass: a band of source rows of one crop
    int32_t crop, row0, nrows;
};
// Row pitch of the horizontal pass's scratch image: rows start 16-byte aligned so the vertical pass reads them as dwords
// (four adjacent output bytes per LDS read) and stages them with 16-byte copies.
__host__ __device__ inline int k1_tmp_pitch(int new_w) { return (new_w * 3 + 15) & ~15; }
// Resampling tables of one crop (resample_tables), at tab + crop.tab_off:
//   width changes:  Taps[new_w] {xmin, n} | pad to 16 | uint4 hk[gh][new_w]   -- coefficients in groups of four taps,
//                   group-major so that adjacent output columns read adjacent 16-byte words; zero beyond tap n
//   height changes: Taps[new_h] | pad to 16 | int vk[new_h][kv]               -- zero beyond tap n
// gh * 4, kv >= 2 * ceil(scale) + 1 = Resample.c's ksize, the upper bound of n.
// groups of four taps per output column of the horizontal pass
__host__ __device__ inline int k1_h_groups(int w, int new_w) { return (2 * ((w + new_w - 1) / new_w) + 1 + 3) >> 2; }
// bytes of LDS the horizontal table of a crop takes beside its band (table padded to whole 1 KiB DMA sweeps)
__host__ __device__ inline int k1_h_table_lds(int w, int new_w) {
    return ((((new_w * 8 + 15) & ~15) + k1_h_groups(w, new_w) * new_w * 16) + 1023) & ~1023;
}
struct K1Layout {
    int gh, kv;
    int64_t hk_off, vt_off, vk_off, bytes;
};
__host__ __device__ inline K1Layout k1_layout(int h, int w, int new_h, int new_w) {
    K1Layout L;
    const bool hp = new_w != w, vp = new_h != h;
    L.gh = hp ? k1_h_groups(w, new_w) : 0;
    L.kv = vp ? ((2 * ((h + new_h - 1) / new_h) + 1 + 3) & ~3) : 0;
    L.hk_off = hp ? (((int64_t)new_w * 8 + 15) & ~(int64_t)15) : 0;
    L.vt_off = L.hk_off + (int64_t)L.gh * new_w * 16;
    L.vk_off = L.vt_off + (vp ? (((int64_t)new_h * 8 + 15) & ~(int64_t)15) : 0);
    L.bytes = (L.vk_off + (int64_t)new_h * L.kv * 4 + 15) & ~(int64_t)15;
    return L;
}
constexpr int K1_H_RPT = 8;       // source rows one item of the horizontal pass filters (bands are whole groups of them)
constexpr int K1_H_RPT_WIDE = 4;  // ... in the second launch (crops whose table + eight rows do not fit the LDS budget)
// both tap tables of every crop, once per crop (f64 on the device exactly as Resample.c computes them on the host)
hipError_t launch_resample_tables(const CropDesc* crops, int n, uint8_t* tab, hipStream_t s);
// horizontal pass over `nwork` bands of ONE class (capi: K1Plan): 0 = eight-row items, table in LDS beside the band;
// 1 = four-row items, table in LDS; 2 = four-row items, table through L1.  lds_bytes = the largest (table +) band.
hipError_t launch_resize_h(const uint8_t* pix, uint8_t* tmp, const CropDesc* crops, const HWork* work, int nwork, int lds_bytes,
                           int cls, const uint8_t* tab, hipStream_t s);
// multi-tile Mllama output: grid_of int32[n,2] (tiles_h, tiles_w); out f32 [n, max_tiles, 3, T, T]
hipError_t launch_resize_v_tiles(const uint8_t* pix, const uint8_t* tmp, const CropDesc* crops, const int32_t* grid_of, int n,
                                 const float* lut, float* out, int T, int max_tiles, hipStream_t s);
// K0: boxes int32[n,4] (x0,y0,x1,y1), offs int64[n] byte offsets into pix; zero fill outside the page
hipError_t launch_crop_boxes(const uint8_t* page, int H, int W, const int32_t* boxes, const int64_t* offs, const HWork* work, int nwork,
                             uint8_t* pix, hipStream_t s);
// bf16(lut[c][u]) == bf16(fma(u, a[c], b[c])) for all 256 u and the 3 channels, VERIFIED on the host when the table was
// built (capi.hip, set_lut): the patch emitter then computes the value instead of reading the table at 64 data-dependent
// LDS addresses per instruction (47 % of that kernel's LDS cycles were bank conflicts, profiles/round2_pmc_k1_sq.csv).
// exact == 0: no such pair was found for this mean / std -- the emitter keeps the table.
struct NormAffine {
    float a[3], b[3];
    int exact;
};
hipError_t launch_resize_v_patchify(const uint8_t* pix, const uint8_t* tmp, const CropDesc* crops, int n,
                                    const float* lut /*[3,256]*/, const NormAffine& aff, void* patches, bool any_resize, const uint8_t* tab, int kv_max,
                                    hipStream_t s);

// K13: class-aware greedy NMS, one workgroup per page (3_combine_grids.py:80-137).  boxes f64[n,4] (x0,y0,x1,y1),
// page p owns boxes [page_offs[p], page_offs[p+1]); order int32[n] is scratch; keep int32[n] receives, per page, the
// page-local indices of the kept boxes in the reference's output order (-1 beyond keep_count[p]).  <= 32768 boxes per page.
hipError_t launch_nms_pages(const double* boxes, const double* scores, const int32_t* classes, const int32_t* page_offs, int pages,
                            double thr, int32_t* order, int32_t* keep, int32_t* keep_count, hipStream_t s);

struct PageSimArgs {
    const void* emb;        // bf16 [N, d]
    int64_t N;
    int d;
    const double* area_pct; // [N]
    const uint8_t* valid;   // [N]
    const int32_t* page_offs; // device [P+1]
    int P;
    const uint8_t* skip;    // [P,P] or null
    int max_query, top_k;
    double max_dist;
    int metric, normalise;
    int64_t pair_lo, pair_hi; // upper-triangle pair ranks [lo, hi) computed by this call (multi-GPU shards); hi < 0 = all
    double* S;              // [P,P]
    // workspace
    float* qsim;            // [nq, N] f32
    const int32_t* qrow;    // device [nq] row index of each query region
    const int32_t* qpage;   // device [nq]
    const int32_t* qstart;  // device [P+1] first query of each page
    int nq;
    void* qemb;             // bf16 [nq, d] gathered query rows
    double* maxbuf;         // [1]
};
hipError_t launch_page_similarity(const PageSimArgs& a, hipStream_t s);

// K12 ranked neighbour lists (neighbours.hip): one wave per row of qsim[nrows, ld] keeps the best `fetch`
// entries in (similarity desc, index asc) order, then filters self / group / score window into top_n
hipError_t launch_topk_rows(const float* qsim, int64_t ld, int N, int nrows, int row0, const int32_t* group, int fetch, int top_n,
                            int keep_self, float min_sim, float max_sim, int32_t* idx_out, float* sim_out, hipStream_t s,
                            const int* run_if = nullptr);
// the same selection over per-row candidate lists (value, column id) produced by the EPI_TOPK GEMM epilogue
hipError_t launch_topk_candidates(const float* cand_val, const int32_t* cand_idx, const int32_t* len, int cap, int nrows, int row0,
                                  const int32_t* group, int fetch, int top_n, int keep_self, float min_sim, float max_sim,
                                  int32_t* idx_out, float* sim_out, hipStream_t s);

// K11 page clustering (cluster.hip): labels_out int32[P], k_out int32[1], scores_out double[16]
hipError_t launch_cluster(const double* S, int P, int n_clusters, int mode, char* ws, int32_t* labels_out, int32_t* k_out,
                          double* scores_out, hipStream_t s);
size_t cluster_workspace_bytes(int P);

// RCCL, resolved at run time (comm.hip)
const char* rccl_ready();  // nullptr = usable, else why not
const char* rccl_error_string(int code);
int rccl_unique_id(void* id128);
int rccl_comm_init(void** comm, int world, const void* id128, int rank);
int rccl_comm_destroy(void* comm);
int rccl_allgather_bytes(const void* send, void* recv, size_t bytes, void* comm, hipStream_t s);

// ---- tile-ViT encoder (tilevit.hip, attention_tiles.hip): Mllama vision tower geometry, fixed at compile time:
// 4 tiles x (1601 -> 1608) tokens, 1280-d, 16 heads of 80, patch 14 on 560 x 560 tiles
hipError_t launch_tile_patchify(const float* pv, void* patches, int64_t npatch, hipStream_t s);
hipError_t launch_tile_assemble(const void* pemb, const float* cls, const float* pre, const float* pos, const float* tilepos, const float* g,
                                const float* b, const int32_t* aid, void* x, int64_t rows, float eps, hipStream_t s);
hipError_t launch_tile_ln_post(void* x, const float* g, const float* b, const float* post, const int32_t* aid, int64_t rows, float eps, hipStream_t s);
hipError_t launch_tile_output(const void* x, const void* inter, int ni, int64_t inter_stride, float* hidden, int64_t out_rows, hipStream_t s);
hipError_t launch_tile_pool(const void* x, const void* inter, int ni, int64_t inter_stride, int n, float* emb_f32, void* emb_bf16, hipStream_t s);
hipError_t launch_attention_tiles(const void* qkv, void* out, const int32_t* ntiles_dev, int n, hipStream_t s, int* guard = nullptr, bool force_redo = false);
