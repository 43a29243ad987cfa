// K11: page clustering -- `cluster_images` of the reference on the GPU.
//
// Restates deprecated_package/weighted_region_clustering.py:452-574 for the code path the
// bundled golden labels pin: scikit-learn >= 1.4 rejects
// `affinity=` and the reference falls back to AgglomerativeClustering(linkage='average')
// over the ROWS of D = 1 - S with the euclidean metric, i.e.
//   scipy pdist('euclidean')  -> scipy linkage 'average' (nearest-neighbour chain, stable
//   sort by height, union-find relabel) -> sklearn _hc_cut (heap of node ids) ->
//   sklearn silhouette_score(metric='precomputed') for k = 2..max_k, strict-> argmax.
// mode 1 runs the same linkage directly on D (the path the reference's first `try:` names).
//
// All arithmetic is f64 with contraction off, in the same operation order as the C/Cython
// the reference executes, so labels and silhouette values reproduce bit for bit.  The
// linkage is inherently sequential (P-1 dependent merges): it runs in ONE workgroup whose
// 1024 lanes parallelise each nearest-neighbour scan and each Lance-Williams row update;
// the O(P^3) row-distance pass before it is an ordinary multi-block kernel.
#include "common.h"
#include "kernels.h"

#pragma clang fp contract(off)

namespace {

constexpr int CT = 1024;       // threads of the linkage workgroup
constexpr int MAXP = 4096;     // pages supported
constexpr int MAX_K_AUTO = 10; // wrc:490

__global__ void to_distance(const double* __restrict__ S, int P, double* __restrict__ D) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < (int64_t)P * P) D[e] = 1.0 - S[e];
}

// Y[i,j] = sqrt(sum_k (D[i,k]-D[j,k])^2), k ascending (scipy euclidean_distance_double)
__global__ __launch_bounds__(256) void pdist_rows(const double* __restrict__ D, int P, double* __restrict__ Y) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = blockIdx.y;
    if (j >= P) return;
    const double* a = D + (int64_t)i * P;
    const double* b = D + (int64_t)j * P;
    double s = 0.0;
    for (int k = 0; k < P; ++k) {
        const double d = a[k] - b[k];
        s += d * d;
    }
    Y[(int64_t)i * P + j] = sqrt(s);
}

__global__ void copy_upper_symmetric(const double* __restrict__ D, int P, double* __restrict__ Y) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (int64_t)P * P) return;
    const int i = (int)(e / P), j = (int)(e - (int64_t)i * P);
    Y[e] = i < j ? D[e] : D[(int64_t)j * P + i];  // squareform -> condensed keeps the upper triangle
}

struct MinPair {
    double d;
    int i;
};

__device__ __forceinline__ MinPair min_pair(MinPair a, MinPair b) {
    return (b.d < a.d || (b.d == a.d && b.i < a.i)) ? b : a;
}

__device__ __forceinline__ MinPair block_min(MinPair v, MinPair* red /*[16]*/) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        MinPair t;
        t.d = __shfl_xor(v.d, o, 64);
        t.i = __shfl_xor(v.i, o, 64);
        v = min_pair(v, t);
    }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    MinPair r = red[0];
#pragma unroll
    for (int k = 1; k < CT / 64; ++k) r = min_pair(r, red[k]);
    return r;
}

// numpy pairwise summation (loops_utils.h.src), recursion unrolled through a template depth
template <int DEPTH>
__device__ double np_pairwise(const double* a, int n) {
    if (n < 8) {
        double r = 0.0;
        for (int i = 0; i < n; ++i) r += a[i];
        return r;
    }
    if (n <= 128) {
        double r[8];
        for (int j = 0; j < 8; ++j) r[j] = a[j];
        int i = 8;
        for (; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; ++j) r[j] += a[i + j];
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; ++i) res += a[i];
        return res;
    }
    int n2 = n / 2;
    n2 -= n2 % 8;
    if constexpr (DEPTH > 0) {
        return np_pairwise<DEPTH - 1>(a, n2) + np_pairwise<DEPTH - 1>(a + n2, n - n2);
    } else {
        return 0.0;  // unreachable for n <= 128 * 2^DEPTH
    }
}

// heapq on an int array (CPython Lib/heapq.py)
__device__ void hq_siftdown(int* h, int startpos, int pos) {
    const int newitem = h[pos];
    while (pos > startpos) {
        const int parentpos = (pos - 1) >> 1;
        const int parent = h[parentpos];
        if (newitem < parent) {
            h[pos] = parent;
            pos = parentpos;
            continue;
        }
        break;
    }
    h[pos] = newitem;
}
__device__ void hq_siftup(int* h, int n, int pos) {
    const int endpos = n, startpos = pos;
    const int newitem = h[pos];
    int childpos = 2 * pos + 1;
    while (childpos < endpos) {
        const int rightpos = childpos + 1;
        if (rightpos < endpos && !(h[childpos] < h[rightpos])) childpos = rightpos;
        h[pos] = h[childpos];
        pos = childpos;
        childpos = 2 * pos + 1;
    }
    h[pos] = newitem;
    hq_siftdown(h, startpos, pos);
}

struct ClusterArgs {
    const double* S;  // [P,P], diagonal already 1
    int P;
    int n_clusters;   // 0 = choose by silhouette
    double* D;        // [P,P]
    double* Y;        // [P,P] working cluster distances
    double* Zh;       // [P] heights (merge order), then sorted
    int32_t* Zx;      // [P] raw merge members
    int32_t* Zy;
    int32_t* child;   // [2*(P-1)] relabelled children, sorted order
    int32_t* parent;  // [2P] tree parents
    int32_t* mark;    // [2P]
    int32_t* lab;     // [P] scratch labels
    double* sil;      // [P]
    int32_t* labels_out;
    int32_t* k_out;
    double* scores_out;  // [16]
};

__global__ __launch_bounds__(CT) void linkage_and_cut(ClusterArgs a) {
    __shared__ int size[MAXP];
    __shared__ int chain[MAXP];
    __shared__ MinPair red[CT / 64];
    __shared__ int heap[MAXP];
    __shared__ int s_misc[8];
    __shared__ double s_freq[MAX_K_AUTO + 1];
    const int tid = threadIdx.x, P = a.P;
    for (int i = tid; i < P; i += CT) size[i] = 1;
    __syncthreads();

    // ---- nearest-neighbour chain (scipy/cluster/_hierarchy.pyx nn_chain) ----
    int clen = 0;
    for (int k = 0; k < P - 1; ++k) {
        if (clen == 0) {
            // first i with size[i] > 0
            MinPair v{INFINITY, 0x7fffffff};
            for (int i = tid; i < P; i += CT)
                if (size[i] > 0) {
                    v = min_pair(v, MinPair{0.0, i});
                    break;
                }
            v = block_min(v, red);
            if (tid == 0) chain[0] = v.i;
            clen = 1;
            __syncthreads();
        }
        int x, y;
        double cur;
        while (true) {
            x = chain[clen - 1];
            int yprev = -1;
            double dprev = INFINITY;
            if (clen > 1) {
                yprev = chain[clen - 2];
                dprev = a.Y[(int64_t)x * P + yprev];
            }
            MinPair v{INFINITY, 0x7fffffff};
            const double* row = a.Y + (int64_t)x * P;
            for (int i = tid; i < P; i += CT)
                if (size[i] != 0 && i != x) v = min_pair(v, MinPair{row[i], i});
            v = block_min(v, red);
            // strict '<' against the previous chain element: it wins every tie
            if (clen > 1 && !(v.d < dprev)) {
                y = yprev;
                cur = dprev;
            } else {
                y = v.i;
                cur = v.d;
            }
            if (clen > 1 && y == yprev) break;
            __syncthreads();
            if (tid == 0) chain[clen] = y;
            ++clen;
            __syncthreads();
        }
        clen -= 2;
        if (x > y) {
            const int t = x;
            x = y;
            y = t;
        }
        const int nx = size[x], ny = size[y];
        __syncthreads();
        if (tid == 0) {
            a.Zx[k] = x;
            a.Zy[k] = y;
            a.Zh[k] = cur;
            size[x] = 0;
            size[y] = nx + ny;
        }
        __syncthreads();
        for (int i = tid; i < P; i += CT) {
            const int ni = size[i];
            if (ni == 0 || i == y) continue;
            const double dxi = a.Y[(int64_t)i * P + x], dyi = a.Y[(int64_t)i * P + y];
            const double nd = ((double)nx * dxi + (double)ny * dyi) / (double)(nx + ny);
            a.Y[(int64_t)i * P + y] = nd;
            a.Y[(int64_t)y * P + i] = nd;
        }
        __threadfence_block();
        __syncthreads();
    }

    // ---- stable sort by height (np.argsort kind='mergesort') as a rank computation ----
    const int M = P - 1;
    int* order = heap;  // reuse: order[rank] = k
    for (int k = tid; k < M; k += CT) {
        const double hk = a.Zh[k];
        int rank = 0;
        for (int m = 0; m < M; ++m) {
            const double hm = a.Zh[m];
            rank += (hm < hk || (hm == hk && m < k)) ? 1 : 0;
        }
        order[rank] = k;
    }
    __syncthreads();
    // ---- union-find relabel (`label`) : serial ----
    if (tid == 0) {
        int* uf = a.mark;  // [2P] scratch as union-find parents
        for (int i = 0; i < 2 * P - 1; ++i) uf[i] = i;
        int next = P;
        for (int r = 0; r < M; ++r) {
            const int k = order[r];
            int xr = a.Zx[k], yr = a.Zy[k];
            int p = xr;
            while (uf[xr] != xr) xr = uf[xr];
            while (uf[p] != xr) {
                const int t = uf[p];
                uf[p] = xr;
                p = t;
            }
            p = yr;
            while (uf[yr] != yr) yr = uf[yr];
            while (uf[p] != yr) {
                const int t = uf[p];
                uf[p] = yr;
                p = t;
            }
            const int lo = xr < yr ? xr : yr, hi = xr < yr ? yr : xr;
            a.child[2 * r] = lo;
            a.child[2 * r + 1] = hi;
            uf[xr] = next;
            uf[yr] = next;
            ++next;
        }
        for (int r = 0; r < M; ++r) {
            a.parent[a.child[2 * r]] = P + r;
            a.parent[a.child[2 * r + 1]] = P + r;
        }
        a.parent[2 * P - 2] = -1;
    }
    __threadfence_block();
    __syncthreads();

    // ---- cut at k clusters (sklearn _hc_cut) + optional silhouette ----
    auto cut = [&](int k, int32_t* out) {
        if (tid == 0) {
            int n = 1;
            const int c0 = a.child[2 * (M - 1)], c1 = a.child[2 * (M - 1) + 1];
            heap[0] = -((c0 > c1 ? c0 : c1) + 1);
            for (int it = 0; it < k - 1; ++it) {
                const int node = -heap[0];
                const int ca = a.child[2 * (node - P)], cb = a.child[2 * (node - P) + 1];
                heap[n] = -ca;  // heappush
                ++n;
                hq_siftdown(heap, 0, n - 1);
                int item = -cb;  // heappushpop
                if (n > 0 && heap[0] < item) {
                    const int t = heap[0];
                    heap[0] = item;
                    item = t;
                    hq_siftup(heap, n, 0);
                }
            }
            for (int i = 0; i < 2 * P - 1; ++i) a.mark[i] = -1;
            for (int i = 0; i < n; ++i) a.mark[-heap[i]] = i;
        }
        __threadfence_block();
        __syncthreads();
        for (int leaf = tid; leaf < P; leaf += CT) {
            int v = leaf;
            while (a.mark[v] < 0) v = a.parent[v];
            out[leaf] = a.mark[v];
        }
        __threadfence_block();
        __syncthreads();
    };

    int best_k = 2;
    if (a.n_clusters > 0) {
        best_k = a.n_clusters;
    } else {
        // wrc:482-490
        if (tid == 0) s_misc[0] = 0;
        __syncthreads();
        int cnt = 0;
        for (int64_t e = tid; e < (int64_t)P * P; e += CT) cnt += a.S[e] > 0.01 ? 1 : 0;
        atomicAdd(&s_misc[0], cnt);
        __syncthreads();
        const int nonzero_pairs = s_misc[0] - P;
        const int max_k = nonzero_pairs < 10 ? (P < 3 ? P : 3) : (P < MAX_K_AUTO ? P : MAX_K_AUTO);
        double best_score = -1.0;
        for (int k = 2; k <= max_k; ++k) {
            cut(k, a.lab);
            double score = NAN;
            if (k < P) {  // sklearn check_number_of_labels: 1 < n_labels < n_samples
                if (tid < k) {
                    int f = 0;
                    for (int j = 0; j < P; ++j) f += a.lab[j] == tid ? 1 : 0;
                    s_freq[tid] = (double)f;
                }
                __syncthreads();
                for (int i = tid; i < P; i += CT) {
                    const double* drow = a.D + (int64_t)i * P;
                    const int own = a.lab[i];
                    double intra = 0.0, inter = INFINITY;
                    for (int c = 0; c < k; ++c) {
                        double s = 0.0;
                        for (int j = 0; j < P; ++j)
                            if (a.lab[j] == c) s += drow[j];
                        if (c == own) {
                            intra = s;
                        } else {
                            const double m = s / s_freq[c];
                            inter = m < inter ? m : inter;
                        }
                    }
                    intra = intra / (s_freq[own] - 1.0);
                    double v = (inter - intra) / (intra > inter ? intra : (inter >= intra ? inter : NAN));
                    if (v != v) v = 0.0;  // np.nan_to_num
                    a.sil[i] = v;
                }
                __threadfence_block();
                __syncthreads();
                if (tid == 0) {
                    const double tot = 0.0 + np_pairwise<6>(a.sil, P);
                    ((double*)s_freq)[MAX_K_AUTO] = tot / (double)P;
                }
                __syncthreads();
                score = s_freq[MAX_K_AUTO];
                if (score > best_score) {
                    best_score = score;
                    best_k = k;
                }
            }
            if (tid == 0) a.scores_out[k] = score;
            __syncthreads();
        }
    }
    cut(best_k, a.labels_out);
    if (tid == 0) a.k_out[0] = best_k;
}

}  // namespace

hipError_t launch_cluster(const double* S, int P, int n_clusters, int mode, char* ws, int32_t* labels_out, int32_t* k_out,
                          double* scores_out, hipStream_t s) {
    if (P < 2 || P > MAXP) return hipErrorInvalidValue;
    ClusterArgs a{};
    a.S = S;
    a.P = P;
    a.n_clusters = n_clusters;
    size_t o = 0;
    auto carve = [&](size_t bytes) {
        char* p = ws + o;
        o += (bytes + 255) & ~(size_t)255;
        return p;
    };
    a.D = (double*)carve((size_t)P * P * 8);
    a.Y = (double*)carve((size_t)P * P * 8);
    a.Zh = (double*)carve((size_t)P * 8);
    a.Zx = (int32_t*)carve((size_t)P * 4);
    a.Zy = (int32_t*)carve((size_t)P * 4);
    a.child = (int32_t*)carve((size_t)P * 8);
    a.parent = (int32_t*)carve((size_t)P * 8);
    a.mark = (int32_t*)carve((size_t)P * 8);
    a.lab = (int32_t*)carve((size_t)P * 4);
    a.sil = (double*)carve((size_t)P * 8);
    a.labels_out = labels_out;
    a.k_out = k_out;
    a.scores_out = scores_out;
    const unsigned nb = (unsigned)(((int64_t)P * P + 255) / 256);
    hipLaunchKernelGGL(to_distance, dim3(nb), dim3(256), 0, s, S, P, a.D);
    if (mode == 0)
        hipLaunchKernelGGL(pdist_rows, dim3((P + 255) / 256, P), dim3(256), 0, s, a.D, P, a.Y);
    else
        hipLaunchKernelGGL(copy_upper_symmetric, dim3(nb), dim3(256), 0, s, a.D, P, a.Y);
    hipLaunchKernelGGL(linkage_and_cut, dim3(1), dim3(CT), 0, s, a);
    return hipGetLastError();
}

size_t cluster_workspace_bytes(int P) {
    const size_t p = (size_t)P;
    return 2 * ((p * p * 8 + 255) & ~(size_t)255) + 12 * ((p * 8 + 255) & ~(size_t)255) + 4096;
}
