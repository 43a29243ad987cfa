// K12: ranked neighbour lists -- the data behind the reference's per-region and
// per-image "similar items" reports.
//
// Restates the selection loop of deprecated_package/region_compare.py:160-353 and
// cross_compare.py:109-235: ask the store for the `fetch` = min(3 top_n, 100) (resp. 5 top_n)
// nearest rows of a query vector -- the query itself is among them -- walk them in ascending
// distance, skip the query (:244), skip rows of the same parent page (:260) and rows outside the
// score window (:269), keep the first top_n (:352).
//
// `topk_rows`: ONE WAVE PER QUERY ROW streams the row of cosine values once (HBM bound: N x 4 B
// per row) and keeps the best `fetch` <= 128 entries as a sorted list spread over the lanes
// (lane l holds positions l and 64 + l), ordered by (similarity descending, index ascending) --
// the order of a stable argsort of the distances.  A value enters the list only if it beats the
// current last entry, which after a short warm-up is rare (about fetch x ln(N / fetch) insertions
// per row), so the scan costs a compare + ballot per 256 values.  Selection then happens on the
// sorted list with ballot prefix ranks.  Bit-exact and order-deterministic: no atomics, no
// floating-point reassociation.
#include <climits>

#include "common.h"
#include "kernels.h"

namespace {

__device__ __forceinline__ bool better(float av, int ai, float bv, int bi) { return av > bv || (av == bv && ai < bi); }

// lane l <- lane l-1 (lane 0 keeps its value): one DPP move instead of a ds_bpermute round trip
__device__ __forceinline__ int wave_shr1(int x) { return __builtin_amdgcn_update_dpp(x, x, 0x138 /* wave_shr:1 */, 0xf, 0xf, false); }
__device__ __forceinline__ float wave_shr1(float x) { return __builtin_bit_cast(float, wave_shr1(__builtin_bit_cast(int, x))); }
__device__ __forceinline__ float lane_of(float x, int l) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), l)); }
__device__ __forceinline__ int lane_of(int x, int l) { return __builtin_amdgcn_readlane(x, l); }

// HI: the list is longer than 64 entries (positions 64.. live in a second register per lane)
// MAPPED: the row is a candidate list (fused path): `len[row]` valid entries whose column ids are
// idx_map[row * ld + position]; ties and the self / group tests use those ids, so the result does not depend
// on the (atomic, arbitrary) order the candidates were appended in.
template <bool HI, bool MAPPED>
__global__ __launch_bounds__(256) void topk_rows(const float* __restrict__ qsim, int64_t ld, int N, int nrows, int row0,
                                                 const int32_t* __restrict__ group, int fetch, int top_n, int keep_self,
                                                 float min_sim, float max_sim, int32_t* __restrict__ idx_out,
                                                 float* __restrict__ sim_out, const int32_t* __restrict__ len,
                                                 const int32_t* __restrict__ idx_map, const int* __restrict__ run_if) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= nrows) return;  // whole wave
    if (run_if && *run_if == 0) return;
    const float* src = qsim + (int64_t)row * ld;
    const int32_t* map = MAPPED ? idx_map + (int64_t)row * ld : nullptr;
    if (MAPPED) N = min(len[row], (int)ld);

    float v_lo = -INFINITY, v_hi = -INFINITY;
    int i_lo = INT_MAX, i_hi = INT_MAX;
    // last list position (fetch - 1) lives in lane tl of the lo or hi register
    const int tl = __builtin_amdgcn_readfirstlane((fetch - 1) & 63);
    float thr_v = -INFINITY;
    int thr_i = INT_MAX;

    auto insert = [&](float cv, int ci) {
        int pos = __popcll(__ballot(better(v_lo, i_lo, cv, ci)));  // the list is sorted: entries 0..pos-1 stay
        if (HI) pos += __popcll(__ballot(better(v_hi, i_hi, cv, ci)));
        const float up_v = wave_shr1(v_lo);
        const int up_i = wave_shr1(i_lo);
        if (HI) {
            const float carry_v = lane_of(v_lo, 63), up_vh = wave_shr1(v_hi);
            const int carry_i = lane_of(i_lo, 63), up_ih = wave_shr1(i_hi);
            const int ph = 64 + lane;
            if (ph == pos) {
                v_hi = cv;
                i_hi = ci;
            } else if (ph > pos) {
                v_hi = lane == 0 ? carry_v : up_vh;
                i_hi = lane == 0 ? carry_i : up_ih;
            }
        }
        if (lane == pos) {
            v_lo = cv;
            i_lo = ci;
        } else if (lane > pos) {
            v_lo = up_v;
            i_lo = up_i;
        }
        thr_v = lane_of(HI ? v_hi : v_lo, tl);
        thr_i = lane_of(HI ? i_hi : i_lo, tl);
    };

    // rows start 16-byte aligned (ld % 4 == 0): 4 consecutive values per lane, 1 KiB per wave step,
    // two steps in flight
    const int steps = (N + 255) / 256;
    auto load_ids = [&](int st) -> int4 {  // column ids of the 4 values of this lane
        const int j = st * 256 + 4 * lane;
        if (!MAPPED) return make_int4(j, j + 1, j + 2, j + 3);
        if (j + 3 < N) return *(const int4*)(map + j);
        return make_int4(j < N ? map[j] : 0, j + 1 < N ? map[j + 1] : 0, j + 2 < N ? map[j + 2] : 0, 0);
    };
    auto load = [&](int st) -> float4 {
        const int j = st * 256 + 4 * lane;
        if (j + 3 < N) return *(const float4*)(src + j);
        float4 r;
        r.x = j < N ? src[j] : -INFINITY;
        r.y = j + 1 < N ? src[j + 1] : -INFINITY;
        r.z = j + 2 < N ? src[j + 2] : -INFINITY;
        r.w = -INFINITY;
        return r;
    };
    float4 cur = steps > 0 ? load(0) : make_float4(0.f, 0.f, 0.f, 0.f);
    int4 cid = steps > 0 ? load_ids(0) : make_int4(0, 0, 0, 0);
    for (int st = 0; st < steps; ++st) {
        const float4 nxt = st + 1 < steps ? load(st + 1) : cur;
        const int4 nid = st + 1 < steps ? load_ids(st + 1) : cid;
        const float e[4] = {cur.x, cur.y, cur.z, cur.w};
        const int id[4] = {cid.x, cid.y, cid.z, cid.w};
        const int j0 = st * 256 + 4 * lane;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            uint64_t m = __ballot(j0 + k < N && better(e[k], id[k], thr_v, thr_i));
            while (m) {
                const int l = __builtin_ctzll(m);
                m &= m - 1;
                const float cv = lane_of(e[k], l);
                const int ci = lane_of(id[k], l);
                if (better(cv, ci, thr_v, thr_i)) insert(cv, ci);  // the bar may have risen meanwhile
            }
        }
        cur = nxt;
        cid = nid;
    }

    // walk the list in order: drop the query, its group, and scores outside the window
    const int self = row0 + row;
    const int gself = group ? group[self] : 0;
    auto keep = [&](float v, int i, int pos) {
        if (pos >= fetch || i == INT_MAX) return false;
        if (!keep_self && i == self) return false;
        if (group && i != self && group[i] == gself) return false;
        return v >= min_sim && v <= max_sim;
    };
    const bool k_lo = keep(v_lo, i_lo, lane), k_hi = keep(v_hi, i_hi, 64 + lane);
    const uint64_t b_lo = __ballot(k_lo), b_hi = __ballot(k_hi);
    const uint64_t below = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    const int r_lo = __popcll(b_lo & below), n_lo = __popcll(b_lo);
    const int r_hi = n_lo + __popcll(b_hi & below);
    int32_t* io = idx_out + (int64_t)row * top_n;
    float* so = sim_out + (int64_t)row * top_n;
    if (k_lo && r_lo < top_n) {
        io[r_lo] = i_lo;
        so[r_lo] = v_lo;
    }
    if (k_hi && r_hi < top_n) {
        io[r_hi] = i_hi;
        so[r_hi] = v_hi;
    }
    const int found = min(top_n, n_lo + __popcll(b_hi));
    for (int k = found + lane; k < top_n; k += 64) {
        io[k] = -1;
        so[k] = 0.f;
    }
}

}  // namespace

hipError_t launch_topk_rows(const float* qsim, int64_t ld, int N, int nrows, int row0, const int32_t* group, int fetch, int top_n,
                            int keep_self, float min_sim, float max_sim, int32_t* idx_out, float* sim_out, hipStream_t s,
                            const int* run_if) {
    if (nrows <= 0) return hipSuccess;
    if (fetch < 1 || fetch > 128 || top_n < 1 || top_n > 128 || (ld & 3) != 0 || N < 1) return hipErrorInvalidValue;
    if (fetch > 64)
        hipLaunchKernelGGL((topk_rows<true, false>), dim3((nrows + 3) / 4), dim3(256), 0, s, qsim, ld, N, nrows, row0, group, fetch, top_n,
                           keep_self, min_sim, max_sim, idx_out, sim_out, nullptr, nullptr, run_if);
    else
        hipLaunchKernelGGL((topk_rows<false, false>), dim3((nrows + 3) / 4), dim3(256), 0, s, qsim, ld, N, nrows, row0, group, fetch, top_n,
                           keep_self, min_sim, max_sim, idx_out, sim_out, nullptr, nullptr, run_if);
    return hipGetLastError();
}

// candidate lists of the fused path: cand_val / cand_idx [nrows, cap], len[nrows] entries appended per row
hipError_t launch_topk_candidates(const float* cand_val, const int32_t* cand_idx, const int32_t* len, int cap, int nrows, int row0,
                                  const int32_t* group, int fetch, int top_n, int keep_self, float min_sim, float max_sim,
                                  int32_t* idx_out, float* sim_out, hipStream_t s) {
    if (nrows <= 0) return hipSuccess;
    if (fetch < 1 || fetch > 128 || top_n < 1 || top_n > 128 || (cap & 3) != 0 || cap < 4) return hipErrorInvalidValue;
    if (fetch > 64)
        hipLaunchKernelGGL((topk_rows<true, true>), dim3((nrows + 3) / 4), dim3(256), 0, s, cand_val, (int64_t)cap, cap, nrows, row0, group, fetch,
                           top_n, keep_self, min_sim, max_sim, idx_out, sim_out, len, cand_idx, nullptr);
    else
        hipLaunchKernelGGL((topk_rows<false, true>), dim3((nrows + 3) / 4), dim3(256), 0, s, cand_val, (int64_t)cap, cap, nrows, row0, group, fetch,
                           top_n, keep_self, min_sim, max_sim, idx_out, sim_out, len, cand_idx, nullptr);
    return hipGetLastError();
}
