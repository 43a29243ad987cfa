// K10: segmented top-k over page segments + area-weighted page-pair reduction.
//
// Restates the page-pair loop of compute_image_similarity_matrix
// (deprecated_package/weighted_region_clustering.py:162-252) on exact brute-force
// similarities (the reference samples them through ChromaDB HNSW queries, wrc:73-95):
//
//   for page i < j (not skipped):                                   wrc:162-192
//     for each of the first `max_query` valid regions r of page i:  wrc:199
//       the min(top_k, |valid regions of j|) nearest rows s of page j,
//       ascending distance, ties by collection order                wrc:207-212
//       keep those with d(r,s) <= max_dist and area_s > 0           wrc:223
//       term = (1 - d) * area_r * area_s   (areas as fractions)     wrc:224-226
//     S[i,j] = S[j,i] = np.sum(terms)                               wrc:231-234
//   off-diagonal /= max off-diagonal ; diagonal = 1                 wrc:246-252
//
// Data flow: (1) pick_queries gathers the query rows (P*max_query slots); (2) the MFMA
// cosine GEMM (gemm.hip, EPI_F32) produces qsim[slot, N]; (3) page_pairs: one 64-lane wave
// per page pair walks contiguous page segments of qsim (regions are grouped by page, so a
// segment is one coalesced stream), selects the top-k by iterated wave-wide lexicographic
// minimum of (distance, index), and sums the terms in f64 in numpy's pairwise order so the
// result is reproducible bit for bit for given similarities; (4) max-normalise.
#include "common.h"
#include "kernels.h"

namespace {

constexpr int MAX_TERMS = 128;  // max_query * top_k <= 128

__global__ void pick_queries(const uint8_t* __restrict__ valid, const int32_t* __restrict__ page_offs, int P, int max_query,
                             int32_t* __restrict__ qrow, int32_t* __restrict__ nvalid) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    int cnt = 0;
    for (int r = page_offs[p]; r < page_offs[p + 1]; ++r) {
        if (valid[r]) {
            if (cnt < max_query) qrow[p * max_query + cnt] = r;
            ++cnt;
        }
    }
    for (int k = cnt; k < max_query; ++k) qrow[p * max_query + k] = -1;
    nvalid[p] = cnt;
}

__global__ __launch_bounds__(256) void gather_rows(const bf16_t* __restrict__ emb, const int32_t* __restrict__ qrow, int d,
                                                   bf16_t* __restrict__ qemb) {
    const int slot = blockIdx.x;
    const int r = qrow[slot];
    for (int c = threadIdx.x * 8; c < d; c += blockDim.x * 8) {
        uint4 v = make_uint4(0, 0, 0, 0);
        if (r >= 0) v = *(const uint4*)(emb + (int64_t)r * d + c);
        *(uint4*)(qemb + (int64_t)slot * d + c) = v;
    }
}

__device__ __forceinline__ double numpy_pairwise_sum(const double* a, int n) {
    // numpy/_core/src/umath/loops_utils.h.src DOUBLE_pairwise_sum, n <= 128; np.sum adds it to 0.
    double res;
    if (n < 8) {
        res = 0.0;
        for (int i = 0; i < n; ++i) res += a[i];
    } else {
        double r[8];
        for (int j = 0; j < 8; ++j) r[j] = a[j];
        int i = 8;
        for (; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; ++j) r[j] += a[i + j];
        res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; ++i) res += a[i];
    }
    return 0.0 + res;
}

// wave-wide maximum: quad and row steps as DPP moves, then the four 16-lane rows through SGPRs
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, x), __builtin_bit_cast(int, x), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float wave_max(float x) {
    x = fmaxf(x, dpp_mov<0xB1>(x));   // quad_perm [1,0,3,2]
    x = fmaxf(x, dpp_mov<0x4E>(x));   // quad_perm [2,3,0,1]
    x = fmaxf(x, dpp_mov<0x124>(x));  // row_ror:4
    x = fmaxf(x, dpp_mov<0x128>(x));  // row_ror:8 -> every lane holds its 16-lane row's maximum
    const int xi = __builtin_bit_cast(int, x);
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(xi, 0)), r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(xi, 16));
    const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(xi, 32)), r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(xi, 48));
    return fmaxf(fmaxf(r0, r1), fmaxf(r2, r3));
}

__global__ __launch_bounds__(256) void page_pairs(PageSimArgs a, const int32_t* __restrict__ qrow_dev,
                                                  const int32_t* __restrict__ nvalid) {
    __shared__ double terms_all[4][MAX_TERMS];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    double* terms = terms_all[wv];
    const int64_t pair = a.pair_lo + (int64_t)blockIdx.x * 4 + wv;
    if (pair >= a.pair_hi) return;
    // unrank pair -> (i, j), i < j, row-major over the upper triangle
    int i = 0;
    {
        // rows before i hold i*(2P-i-1)/2 pairs
        double fi = ((2.0 * a.P - 1.0) - sqrt((2.0 * a.P - 1.0) * (2.0 * a.P - 1.0) - 8.0 * (double)pair)) * 0.5;
        i = (int)fi;
        while ((int64_t)i * (2 * a.P - i - 1) / 2 > pair) --i;
        while ((int64_t)(i + 1) * (2 * a.P - i - 2) / 2 <= pair) ++i;
    }
    const int j = (int)(pair - (int64_t)i * (2 * a.P - i - 1) / 2) + i + 1;

    double result = 0.0;
    const bool skipped = (a.skip && a.skip[(int64_t)i * a.P + j]) || nvalid[i] == 0 || nvalid[j] == 0;
    if (!skipped) {
        const int seg0 = a.page_offs[j], L = a.page_offs[j + 1] - seg0;
        const int n_results = min(a.top_k, nvalid[j]);
        const int nq = min(a.max_query, nvalid[i]);
        int nterms = 0;
        for (int qi = 0; qi < nq; ++qi) {
            const int slot = i * a.max_query + qi;
            const int r = qrow_dev[slot];
            const double area_i = a.area_pct[r] / 100.0;
            if (area_i == 0.0) continue;  // wrc:203
            const float* srow = a.qsim + (int64_t)slot * a.N + seg0;
            if (L <= 256) {
                // Page segments of up to 256 regions live in four registers per lane (c = lane + 64 t).  d is a
                // strictly decreasing, exact function of the f32 similarity, so "ascending (d, index)" is
                // "descending similarity, ascending index": k rounds of wave-max, lowest holder, knock out.
                float v0 = lane < L ? srow[lane] : -INFINITY, v1 = lane + 64 < L ? srow[lane + 64] : -INFINITY;
                float v2 = lane + 128 < L ? srow[lane + 128] : -INFINITY, v3 = lane + 192 < L ? srow[lane + 192] : -INFINITY;
                for (int k = 0; k < n_results; ++k) {
                    const float M = wave_max(fmaxf(fmaxf(v0, v1), fmaxf(v2, v3)));
                    if (M == -INFINITY) break;  // segment exhausted
                    int tsel = 0;
                    uint64_t bsel = __ballot(v0 == M);
                    if (!bsel) { bsel = __ballot(v1 == M); tsel = 1; }
                    if (!bsel) { bsel = __ballot(v2 == M); tsel = 2; }
                    if (!bsel) { bsel = __ballot(v3 == M); tsel = 3; }
                    const int lsel = __builtin_ctzll(bsel);
                    const int best_idx = lsel + 64 * tsel;
                    if (lane == lsel) {
                        if (tsel == 0) v0 = -INFINITY;
                        else if (tsel == 1) v1 = -INFINITY;
                        else if (tsel == 2) v2 = -INFINITY;
                        else v3 = -INFINITY;
                    }
                    const double best_d = a.metric == 0 ? 1.0 - (double)M : 2.0 - 2.0 * (double)M;
                    const double area_j = a.area_pct[seg0 + best_idx] / 100.0;
                    if (best_d <= a.max_dist && area_j > 0.0) {
                        if (lane == 0 && nterms < MAX_TERMS) terms[nterms] = (1.0 - best_d) * area_i * area_j;
                        ++nterms;
                    }
                }
                continue;
            }
            double last_d = -INFINITY;
            int last_idx = -1;
            for (int k = 0; k < n_results; ++k) {
                // lexicographic minimum of (d, idx) strictly greater than (last_d, last_idx)
                double best_d = INFINITY;
                int best_idx = 0x7fffffff;
                for (int c = lane; c < L; c += 64) {
                    const double sim = (double)srow[c];
                    const double dd = a.metric == 0 ? 1.0 - sim : 2.0 - 2.0 * sim;
                    const bool after = (dd > last_d) || (dd == last_d && c > last_idx);
                    const bool better = (dd < best_d) || (dd == best_d && c < best_idx);
                    if (after && better) {
                        best_d = dd;
                        best_idx = c;
                    }
                }
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) {
                    const double od = __shfl_xor(best_d, o, 64);
                    const int oi = __shfl_xor(best_idx, o, 64);
                    if (od < best_d || (od == best_d && oi < best_idx)) {
                        best_d = od;
                        best_idx = oi;
                    }
                }
                if (best_idx == 0x7fffffff) break;  // segment exhausted
                last_d = best_d;
                last_idx = best_idx;
                const double area_j = a.area_pct[seg0 + best_idx] / 100.0;
                if (best_d <= a.max_dist && area_j > 0.0) {
                    if (lane == 0 && nterms < MAX_TERMS) terms[nterms] = (1.0 - best_d) * area_i * area_j;
                    ++nterms;
                }
            }
        }
        if (nterms > 0) {
            __builtin_amdgcn_wave_barrier();
            if (lane == 0) result = numpy_pairwise_sum(terms, min(nterms, MAX_TERMS));
            result = __shfl(result, 0, 64);
        }
    }
    if (lane == 0) {
        a.S[(int64_t)i * a.P + j] = result;
        a.S[(int64_t)j * a.P + i] = result;
    }
}

__global__ __launch_bounds__(1024) void offdiag_max(const double* __restrict__ S, int P, double* __restrict__ out) {
    __shared__ double red[16];
    double m = 0.0;  // np.max(S - diag) over a matrix whose diagonal entries become 0
    const int64_t n = (int64_t)P * P;
    for (int64_t e = threadIdx.x; e < n; e += 1024) {
        const int r = (int)(e / P), c = (int)(e - (int64_t)r * P);
        const double v = (r == c) ? 0.0 : S[e];
        m = v > m ? v : m;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const double t = __shfl_xor(m, o, 64);
        m = t > m ? t : m;
    }
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < 16; ++k) m = red[k] > m ? red[k] : m;
        out[0] = m;
    }
}

__global__ void normalise_S(double* __restrict__ S, int P, const double* __restrict__ mx, int zero_diag_first) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (int64_t)P * P) return;
    const int r = (int)(e / P), c = (int)(e - (int64_t)r * P);
    if (r == c) {
        S[e] = zero_diag_first ? 0.0 : 1.0;
    } else if (!zero_diag_first && mx[0] > 0.0) {
        S[e] = S[e] / mx[0];
    }
}

}  // namespace

hipError_t launch_page_similarity(const PageSimArgs& a, hipStream_t s) {
    if (a.P <= 0) return hipSuccess;
    if (a.max_query * a.top_k > MAX_TERMS) return hipErrorInvalidValue;
    int32_t* qrow = (int32_t*)a.qrow;
    int32_t* nvalid = (int32_t*)a.qpage;  // reused as the per-page valid-region count
    hipLaunchKernelGGL(pick_queries, dim3((a.P + 63) / 64), dim3(64), 0, s, a.valid, a.page_offs, a.P, a.max_query, qrow, nvalid);
    const int slots = a.P * a.max_query;
    hipLaunchKernelGGL(gather_rows, dim3(slots), dim3(128), 0, s, (const bf16_t*)a.emb, qrow, a.d, (bf16_t*)a.qemb);
    GemmArgs g{};
    g.A = a.qemb;
    g.W = a.emb;
    g.M = slots;
    g.N = (int)a.N;
    g.K = a.d;
    g.outf = a.qsim;
    g.ldf = a.N;
    hipError_t e = launch_gemm(EPI_F32, g, s);
    if (e != hipSuccess) return e;
    // S starts as np.zeros (wrc:160); this call fills the pairs of its rank range [lo, hi) (all of them
    // unless a multi-GPU caller shards the upper triangle) and leaves the rest 0, so shards add up exactly
    const int64_t npairs = (int64_t)a.P * (a.P - 1) / 2;
    PageSimArgs b = a;
    if (b.pair_hi < 0 || b.pair_hi > npairs) b.pair_hi = npairs;
    if (b.pair_lo < 0) b.pair_lo = 0;
    e = hipMemsetAsync(a.S, 0, (size_t)a.P * a.P * sizeof(double), s);
    if (e != hipSuccess) return e;
    if (b.pair_hi > b.pair_lo)
        hipLaunchKernelGGL(page_pairs, dim3((unsigned)((b.pair_hi - b.pair_lo + 3) / 4)), dim3(256), 0, s, b, qrow, nvalid);
    if (a.normalise) {
        hipLaunchKernelGGL(offdiag_max, dim3(1), dim3(1024), 0, s, a.S, a.P, a.maxbuf);
        hipLaunchKernelGGL(normalise_S, dim3((unsigned)(((int64_t)a.P * a.P + 255) / 256)), dim3(256), 0, s, a.S, a.P, a.maxbuf, 0);
    }
    return hipGetLastError();
}
