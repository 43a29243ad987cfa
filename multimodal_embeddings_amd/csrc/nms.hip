// K13: class-aware greedy non-maximum suppression of the detections of many pages at once.
//
// Restates apply_non_max_suppression + calculate_iou of the step that merges the detector's grid passes into the
// boxes of a page (3_combine_grids.py:44-137): repeatedly keep the highest-scoring box that is left
// (the FIRST of equal scores) and drop every remaining box of the same class whose IoU with it exceeds the threshold.
// That is a greedy pass over the boxes in stable descending-score order; the reference does it with list.index /
// list.pop in O(n^2) Python per page.  Here one workgroup owns one page:
//   1. rank of box i = #{j : s_j > s_i or (s_j == s_i and j < i)}  (all pairs, parallel) -> order[rank] = i;
//   2. walk the order; a surviving box is kept and all threads test it against the later boxes (float64, the
//      reference's operation order, contraction off), marking the suppressed ones in an LDS bit set.
// Latency-bound integer / f64 work, no MFMA; pages are independent, so a corpus of pages fills the chip.
#include "common.h"
#include "kernels.h"

#pragma clang fp contract(off)

namespace {

constexpr int NMS_MAX_BOXES = 32768;  // per page (LDS bit set of 1024 words)

// calculate_iou(box1 = the box just kept, box2) -- 3_combine_grids.py:44-78
__device__ __forceinline__ double iou_ref(const double4 a, const double4 b) {
    const double xl = fmax(a.x, b.x), yt = fmax(a.y, b.y), xr = fmin(a.z, b.z), yb = fmin(a.w, b.w);
    if (xr < xl || yb < yt) return 0.0;
    const double inter = (xr - xl) * (yb - yt);
    const double a1 = (a.z - a.x) * (a.w - a.y), a2 = (b.z - b.x) * (b.w - b.y);
    const double uni = (a1 + a2) - inter;
    return uni > 0 ? inter / uni : 0.0;
}

__global__ __launch_bounds__(256) void nms_pages(const double* __restrict__ boxes, const double* __restrict__ scores,
                                                 const int32_t* __restrict__ classes, const int32_t* __restrict__ page_offs,
                                                 double thr, int32_t* __restrict__ order, int32_t* __restrict__ keep,
                                                 int32_t* __restrict__ keep_count) {
    __shared__ uint32_t removed[NMS_MAX_BOXES / 32];
    const int page = blockIdx.x, tid = threadIdx.x;
    const int base = page_offs[page], n = page_offs[page + 1] - base;
    const double4* bx = (const double4*)boxes + base;
    const double* sc = scores + base;
    const int32_t* cl = classes + base;
    int32_t* ord = order + base;
    int32_t* kp = keep + base;
    for (int i = tid; i < (n + 31) / 32; i += 256) removed[i] = 0;
    // The rank must be a TOTAL order or order[] is not a permutation and the walk below indexes with stale words: a NaN
    // score compares false with everything, so it is ranked as -inf here (mme_nms_boxes rejects NaN scores before the
    // launch; this keeps the kernel in bounds whatever it is handed).
    for (int i = tid; i < n; i += 256) {
        const double si = sc[i] == sc[i] ? sc[i] : -INFINITY;
        int r = 0;
        for (int j = 0; j < n; ++j) {
            const double sj = sc[j] == sc[j] ? sc[j] : -INFINITY;
            r += (sj > si || (sj == si && j < i)) ? 1 : 0;
        }
        ord[r] = i;
        kp[i] = -1;
    }
    __syncthreads();  // (global writes of this workgroup are visible to it after the barrier's release / acquire)
    __threadfence_block();
    int cnt = 0;
    for (int r = 0; r < n; ++r) {
        const int i = ord[r];  // uniform
        if ((removed[i >> 5] >> (i & 31)) & 1) continue;  // uniform: same LDS word for every thread
        if (tid == 0) kp[cnt] = i;
        ++cnt;
        const double4 cur = bx[i];
        const int ci = cl[i];
        for (int r2 = r + 1 + tid; r2 < n; r2 += 256) {
            const int j = ord[r2];
            if ((removed[j >> 5] >> (j & 31)) & 1) continue;
            if (cl[j] == ci && iou_ref(cur, bx[j]) > thr) atomicOr(&removed[j >> 5], 1u << (j & 31));
        }
        __syncthreads();
    }
    if (tid == 0) keep_count[page] = cnt;
}

}  // namespace

hipError_t launch_nms_pages(const double* boxes, const double* scores, const int32_t* classes, const int32_t* page_offs, int pages,
                            double thr, int32_t* order, int32_t* keep, int32_t* keep_count, hipStream_t s) {
    if (pages <= 0) return hipSuccess;
    hipLaunchKernelGGL(nms_pages, dim3(pages), dim3(256), 0, s, boxes, scores, classes, page_offs, thr, order, keep, keep_count);
    return hipGetLastError();
}
