// K1: batched variable-size crop -> Pillow-BILINEAR fit to 224 -> zero pad -> normalise ->
// patchify (bf16 patch matrix [n*196, 768] in conv order (c, ky, kx)).
//
// Restates, bit for bit on the uint8 side, what every crop goes through in the reference
// (deprecated_package/embedder.py:117-121 -> transformers image_processing_pil_mllama.py:
// 483-541 -> Pillow libImaging/Resample.c, 8-bit path): separable triangle filter whose
// support grows with the down-scale factor, 22-bit fixed-point coefficients, horizontal pass
// then vertical pass with uint8 rounding after each, pad BEFORE normalisation.
//
// HBM-bound byte work (no MFMA): reads are 16-byte coalesced row streams staged through LDS,
// the patch matrix is written as whole 1536-byte patch rows.
//   resample_tables   one workgroup per crop: both tap tables (window + coefficients per output coordinate), f64 as
//                     Resample.c, laid out for the two passes (K1Layout, kernels.h).
//   resize_h          one workgroup per (crop, band of source rows): band -> LDS with 16-byte loads; a thread filters
//                     one output pixel column of FOUR rows: per group of four taps one 16-byte coefficient word (L1 / L2)
//                     and one 12-byte LDS read per row (4 pixels x RGB), 48 24-bit multiply-adds; the three result bytes
//                     of four neighbouring lanes are exchanged inside the quad (DPP) and leave as dword stores into a
//                     scratch image whose rows are 16-byte aligned.
//   resize_v_patchify one workgroup per (crop, patch row): the source-row window of the 16-row canvas band goes through
//                     an LDS window in chunks; a thread owns four adjacent canvas bytes of one row (one dword LDS read
//                     per tap, accumulators in registers across chunks); then LUT-normalise and emit 14 patches.
//
// The fixed-point sums are exact in 32-bit unsigned arithmetic: triangle weights are >= 0 and sum to 2^22 +- n/2, so
// 255 * sum + 2^21 < 2^31, and v_mad_u32_u24 multiplies an 8-bit pixel by a < 2^24 coefficient exactly.
// Contraction is disabled so no fused multiply-add changes a rounding of the f64 coefficient arithmetic.
#include <cstdlib>

#include "common.h"
#include "kernels.h"

#pragma clang fp contract(off)

namespace {

constexpr int PRECISION_BITS = 32 - 8 - 2;
constexpr int MAX_TAPS = 160;      // window <= 2*ceil(scale)+1 and scale < 2*MAX_DIM/224

struct Taps {
    int xmin, n;
};
struct __attribute__((packed)) Pix12 {  // 4 RGB pixels at ANY byte address (gfx950 reads unaligned LDS words)
    uint32_t a, b, c;
};

// One output coordinate's window and fixed-point weights (Resample.c precompute_coeffs +
// normalize_coeffs_8bpc, bilinear filter, box = whole image).  store(i, k) receives tap i's coefficient.
// [xmin, xmax) of one output coordinate (the first lines of precompute_coeffs)
__device__ __forceinline__ Taps taps_window(int in_size, int out_size, int xx) {
    const double scale = (double)in_size / (double)out_size;
    const double filterscale = scale < 1.0 ? 1.0 : scale;
    const double support = 1.0 * filterscale;
    const double center = (xx + 0.5) * scale;
    int xmin = (int)(center - support + 0.5);
    if (xmin < 0) xmin = 0;
    int xmax = (int)(center + support + 0.5);
    if (xmax > in_size) xmax = in_size;
    return Taps{xmin, xmax - xmin};
}
template <class Store>
__device__ __forceinline__ Taps compute_taps_to(int in_size, int out_size, int xx, Store store) {
    const double scale = (double)in_size / (double)out_size;
    const double filterscale = scale < 1.0 ? 1.0 : scale;
    const double ss = 1.0 / filterscale;
    const double center = (xx + 0.5) * scale;
    const Taps win = taps_window(in_size, out_size, xx);
    const int xmin = win.xmin, n = win.n;
    double ww = 0.0;
    for (int x = 0; x < n; ++x) {
        double t = (x + xmin - center + 0.5) * ss;
        if (t < 0.0) t = -t;
        const double w = t < 1.0 ? 1.0 - t : 0.0;
        ww += w;
    }
    for (int x = 0; x < n; ++x) {
        double t = (x + xmin - center + 0.5) * ss;
        if (t < 0.0) t = -t;
        double w = t < 1.0 ? 1.0 - t : 0.0;
        if (ww != 0.0) w /= ww;
        store(x, w < 0 ? (int)(-0.5 + w * (double)(1 << PRECISION_BITS)) : (int)(0.5 + w * (double)(1 << PRECISION_BITS)));
    }
    return Taps{xmin, n};
}
__device__ __forceinline__ Taps compute_taps(int in_size, int out_size, int xx, int* kk /*[MAX_TAPS]*/) {
    return compute_taps_to(in_size, out_size, xx, [&](int i, int k) { kk[i] = k; });
}

__device__ __forceinline__ uint8_t clip8(int v) {
    v >>= PRECISION_BITS;
    return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
}
__device__ __forceinline__ uint32_t clip8u(uint32_t v) {
    v >>= PRECISION_BITS;
    return v > 255u ? 255u : v;
}
__device__ __forceinline__ uint32_t mad24(uint32_t a, uint32_t b, uint32_t c) { return __umul24(a, b) + c; }

// Contiguous 16-byte-aligned global range -> LDS by LDS-DMA: every wave instruction moves 64 x 16 bytes to
// (wave-uniform base) + lane * 16 with nothing staged in registers, all requests in flight at once (a register-staged
// copy loop waits for each load before it stores: one global latency per 4 KiB).  Lanes past the end re-read the last
// vector into up to 1008 bytes of slack behind the range, which the caller's LDS allocation includes (DMA_SLACK).
constexpr int DMA_SLACK = 1024;
template <int NT = 256>
__device__ __forceinline__ void dma_range_to_lds(const uint4* __restrict__ g, char* lds, int nvec, int tid) {
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    for (int i = wave * 64; i < nvec; i += NT) glds16(g + min(i + lane, nvec - 1), lds + (size_t)i * 16);
}
__device__ __forceinline__ void dma_wait_all() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// Both tap tables of a crop, ONCE per crop (a band / patch-row workgroup used to recompute its own: f64 loops over up to
// 2 * scale + 1 taps per output coordinate, ~3 k cycles, more than filtering a small band costs).
__global__ __launch_bounds__(256) void resample_tables(const CropDesc* __restrict__ crops, uint8_t* __restrict__ tab) {
    const CropDesc c = crops[blockIdx.x];
    const K1Layout L = k1_layout(c.h, c.w, c.new_h, c.new_w);
    uint8_t* base = tab + c.tab_off;
    if (c.new_w != c.w) {
        Taps* taps = (Taps*)base;
        int* hk = (int*)(base + L.hk_off);
        for (int x = threadIdx.x; x < c.new_w; x += 256) {
            const Taps t = compute_taps_to(c.w, c.new_w, x, [&](int i, int k) { hk[((int64_t)(i >> 2) * c.new_w + x) * 4 + (i & 3)] = k; });
            for (int i = t.n; i < L.gh * 4; ++i) hk[((int64_t)(i >> 2) * c.new_w + x) * 4 + (i & 3)] = 0;
            taps[x] = t;
        }
    }
    if (c.new_h != c.h) {
        Taps* taps = (Taps*)(base + L.vt_off);
        int* vk = (int*)(base + L.vk_off);
        for (int y = threadIdx.x; y < c.new_h; y += 256) {
            int* row = vk + (int64_t)y * L.kv;
            const Taps t = compute_taps_to(c.h, c.new_h, y, [&](int i, int k) { row[i] = k; });
            for (int i = t.n; i < L.kv; ++i) row[i] = 0;
            taps[y] = t;
        }
    }
}

// Horizontal pass.  One workgroup = one band of source rows of one crop (a whole number of K1_H_RPT-row groups except at
// the crop's end); the band is a single contiguous byte range, fetched by LDS-DMA (all requests in flight at once).
// Work item = (row group, output column): lanes of a quad are four neighbouring columns.
// TAB_LDS: the crop's window + coefficient table rides into LDS with the band, so the item loop holds NO vector-memory
// load: gfx950's vmcnt counts stores too, and with table reads in the loop every item waited for the previous item's
// stores to complete (one write latency per item).  TAB_LDS = false (very wide crops, whose table does not fit beside
// four source rows) keeps the table in L1 / L2.
// RPT = source rows one item filters: 8 where the table rides in LDS (the per-item set-up -- window, pointers, packing -- is
// amortised over twice the multiply-adds; the kernel is vector-ALU bound and two thirds of its instructions were not
// multiply-adds), 4 for the very wide crops of the second launch (eight of their rows would not fit the LDS).
template <bool TAB_LDS, int RPT>
__global__ __launch_bounds__(256) void resize_h(const uint8_t* __restrict__ pix, uint8_t* __restrict__ tmp,
                                                const CropDesc* __restrict__ crops, const HWork* __restrict__ work,
                                                const uint8_t* __restrict__ tab) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const HWork wk = work[blockIdx.x];
    const CropDesc c = crops[wk.crop];
    const int tid = threadIdx.x;
    const int row_bytes = c.w * 3;
    const int hk_off = (c.new_w * 8 + 15) & ~15;
    const uint8_t* gtab = tab + c.tab_off;
    // LDS: [table, padded to whole 1 KiB DMA sweeps] | band (+ slack)
    const int tab_bytes = TAB_LDS ? hk_off + k1_h_groups(c.w, c.new_w) * c.new_w * 16 : 0;
    const int tab_pad = (tab_bytes + DMA_SLACK - 1) & ~(DMA_SLACK - 1);
    if (TAB_LDS) dma_range_to_lds((const uint4*)gtab, smem, tab_bytes >> 4, tid);
    uint8_t* band = (uint8_t*)smem + tab_pad;
    const uint8_t* src = pix + c.src_off + (int64_t)wk.row0 * row_bytes;
    const int nbytes = wk.nrows * row_bytes;
    const uintptr_t a0 = (uintptr_t)src & ~(uintptr_t)15;
    const int lead = (int)((uintptr_t)src - a0);
    const int nvec = (lead + nbytes + 15) >> 4;
    dma_range_to_lds((const uint4*)a0, (char*)band, nvec, tid);
    const int xw = (c.new_w + 3) & ~3;  // columns rounded up to whole quads (the extra lanes repeat the last column)
    const int nrg = (wk.nrows + RPT - 1) / RPT;
    const int nitems = nrg * xw;
    const uint8_t* bb = band + lead;
    const int pitch = k1_tmp_pitch(c.new_w);
    uint8_t* dst = tmp + c.tmp_off + (int64_t)wk.row0 * pitch;  // wave-uniform base; lane offsets below stay 32-bit
    const int j = tid & 3;  // position in the quad (256 and xw are multiples of 4: quads never straddle items' rows)
    // lanes j = 0..2 of a quad write the quad's 12 output bytes as three dwords: dword j = (v_j >> 8j) | (v_{j+1} << (24 - 8j))
    const int sh_own = 8 * j, sh_nb = 24 - 8 * j;
    auto item = [&](int e, Taps t, const uint4* __restrict__ kcol /* this column's coefficient groups, stride new_w */, uint4 k) {
        const int rg = e / xw, xq = e - rg * xw;
        const int y0 = rg * RPT;
        const int ng = (t.n + 3) >> 2;
        const uint8_t* p[RPT];
#pragma unroll
        for (int r = 0; r < RPT; ++r) p[r] = bb + min(y0 + r, wk.nrows - 1) * row_bytes + t.xmin * 3;
        uint32_t acc[RPT][3];
#pragma unroll
        for (int r = 0; r < RPT; ++r) acc[r][0] = acc[r][1] = acc[r][2] = 1u << (PRECISION_BITS - 1);
        for (int g = 0; g < ng; ++g) {
            const uint4 kn = g + 1 < ng ? kcol[(int64_t)(g + 1) * c.new_w] : uint4{0, 0, 0, 0};
#pragma unroll
            for (int r = 0; r < RPT; ++r) {
                const Pix12 d = *(const Pix12*)(p[r] + g * 12);
                acc[r][0] = mad24(d.a & 0xff, k.x, acc[r][0]);
                acc[r][1] = mad24((d.a >> 8) & 0xff, k.x, acc[r][1]);
                acc[r][2] = mad24((d.a >> 16) & 0xff, k.x, acc[r][2]);
                acc[r][0] = mad24(d.a >> 24, k.y, acc[r][0]);
                acc[r][1] = mad24(d.b & 0xff, k.y, acc[r][1]);
                acc[r][2] = mad24((d.b >> 8) & 0xff, k.y, acc[r][2]);
                acc[r][0] = mad24((d.b >> 16) & 0xff, k.z, acc[r][0]);
                acc[r][1] = mad24(d.b >> 24, k.z, acc[r][1]);
                acc[r][2] = mad24(d.c & 0xff, k.z, acc[r][2]);
                acc[r][0] = mad24((d.c >> 8) & 0xff, k.w, acc[r][0]);
                acc[r][1] = mad24((d.c >> 16) & 0xff, k.w, acc[r][1]);
                acc[r][2] = mad24(d.c >> 24, k.w, acc[r][2]);
            }
            k = kn;
        }
        uint32_t out[RPT];
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
            const uint32_t v = clip8u(acc[r][0]) | (clip8u(acc[r][1]) << 8) | (clip8u(acc[r][2]) << 16);
            const uint32_t nb = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xF9 /* quad_perm [1,2,3,3] */, 0xF, 0xF, true);
            out[r] = (v >> sh_own) | (nb << sh_nb);  // (j = 3: a value nobody stores)
        }
        const uint32_t o = (uint32_t)((xq & ~3) * 3 + 4 * j);
        if (j < 3 && o < (uint32_t)(c.new_w * 3)) {
            const uint32_t off0 = (uint32_t)y0 * (uint32_t)pitch + o;
            if (y0 + RPT <= wk.nrows) {  // wave-uniform in all but a crop's last row group
#pragma unroll
                for (int r = 0; r < RPT; ++r) *(uint32_t*)(dst + (off0 + (uint32_t)(r * pitch))) = out[r];
            } else {
#pragma unroll
                for (int r = 0; r < RPT; ++r)
                    if (y0 + r < wk.nrows) *(uint32_t*)(dst + (off0 + (uint32_t)(r * pitch))) = out[r];
            }
        }
    };
    if constexpr (TAB_LDS) {
        dma_wait_all();
        __syncthreads();
        const Taps* taps = (const Taps*)smem;
        const uint4* hk = (const uint4*)(smem + hk_off);
        for (int e = tid; e < nitems; e += 256) {
            const int xx = min(e % xw, c.new_w - 1);
            item(e, taps[xx], hk + xx, hk[xx]);
        }
    } else {
        const Taps* __restrict__ taps = (const Taps*)gtab;
        const uint4* __restrict__ hk = (const uint4*)(gtab + hk_off);
        // the first item's window and coefficients travel while the band lands; later ones one item ahead
        int e = tid;
        int xx = min(e % xw, c.new_w - 1);
        Taps t = e < nitems ? taps[xx] : Taps{0, 0};
        uint4 k0 = e < nitems ? hk[xx] : uint4{0, 0, 0, 0};
        dma_wait_all();
        __syncthreads();
        while (e < nitems) {
            const int e_n = e + 256;
            const int xx_n = min(e_n % xw, c.new_w - 1);
            Taps t_n = Taps{0, 0};
            uint4 k_n = uint4{0, 0, 0, 0};
            if (e_n < nitems) {
                t_n = taps[xx_n];
                k_n = hk[xx_n];
            }
            item(e, t, hk + xx, k0);
            e = e_n;
            xx = xx_n;
            t = t_n;
            k0 = k_n;
        }
    }
}

// Vertical pass + zero pad + normalise + patchify.  Three forms of the canvas fill:
//   (a) unchanged 224-wide rows (the synthetic 224 x 224 workload): one contiguous 10752-byte band, 16-byte copies;
//   (b) the source is the horizontal pass's scratch image (16-byte aligned rows): dword path below;
//   (c) the width was not resized (rows of the packed crop at any alignment; rare): byte reads straight from global.
// RESIZE = false: the instantiation for batches of 224 x 224 crops only (form (b) compiled out, 256 threads: the pure
// stream keeps its occupancy).  RESIZE = true: 512 threads share one canvas band and window, so the LDS-bound three or
// four workgroups per CU still are 24-32 waves.
// Dynamic LDS: kk[16][kvs] (this band's coefficient rows; kvs = the batch's largest K1Layout::kv) | window.
template <bool RESIZE>
__global__ __launch_bounds__(RESIZE ? 512 : 256, RESIZE ? 8 : 1) void resize_v_patchify(const uint8_t* __restrict__ pix, const uint8_t* __restrict__ tmp,
                                                         const CropDesc* __restrict__ crops, const float* __restrict__ lut,
                                                         bf16_t* __restrict__ patches, const uint8_t* __restrict__ tab, int window_bytes,
                                                         int kvs, const NormAffine aff) {
    constexpr int NT = RESIZE ? 512 : 256;
    constexpr int ROW = VIT_IMG * 3, ROW4 = ROW / 4, NIT = (VIT_PATCH * ROW4 + NT - 1) / NT;
    __shared__ __attribute__((aligned(16))) uint8_t canvas[VIT_PATCH * ROW + 16];
    __shared__ Taps taps[VIT_PATCH];
    __shared__ float slut[3 * 256];
    extern __shared__ __attribute__((aligned(16))) char dyn[];
    int* kk = (int*)dyn;
    uint8_t* window = (uint8_t*)dyn + (size_t)VIT_PATCH * kvs * sizeof(int);  // source rows feeding this band
    const int crop = blockIdx.x / VIT_GRID, py = blockIdx.x - crop * VIT_GRID;
    const CropDesc c = crops[crop];
    const int tid = threadIdx.x;
    const bool hpass = c.new_w != c.w;       // Resample.c: horizontal pass only when width changes
    const bool vpass = c.new_h != c.h;
    const int src_row_bytes = c.new_w * 3;   // after the horizontal pass (or unchanged width)
    const int y_first = py * VIT_PATCH, nout = min(y_first + VIT_PATCH, c.new_h) - y_first;  // canvas rows with pixels (may be <= 0)
    const uint8_t* src = hpass ? tmp + c.tmp_off : pix + c.src_off;
    const uint8_t* band = src + (int64_t)y_first * src_row_bytes;
    const bool form_a = !vpass && src_row_bytes == ROW && nout == VIT_PATCH && (((uintptr_t)band) & 15) == 0;
    const bool form_b = RESIZE && !form_a && hpass && nout > 0 && window_bytes > 0;
    // form (b): source rows [r0, r1) feed this band -- known from the two window formulas alone, so the first chunk's DMA
    // starts before the coefficient rows are fetched
    int r0 = 0, r1 = 0, rows_chunk = 1;
    const int pitch = k1_tmp_pitch(c.new_w), pitch4 = pitch >> 2;
    if (form_b) {
        if (vpass) {
            const Taps a = taps_window(c.h, c.new_h, y_first), z = taps_window(c.h, c.new_h, y_first + nout - 1);
            r0 = a.xmin;
            r1 = z.xmin + z.n;
        } else {
            r0 = y_first;
            r1 = y_first + nout;
        }
        rows_chunk = max(window_bytes / pitch, 1);
        dma_range_to_lds<NT>((const uint4*)(src + (int64_t)r0 * pitch), (char*)window, (min(r0 + rows_chunk, r1) - r0) * (pitch >> 4), tid);
    }
    const bool affine = RESIZE && aff.exact;  // the all-224 x 224 instantiation is an HBM-bound stream: its table form measured 5 % faster (tools/k1_ab.py)
    if (!affine)
        for (int i = tid; i < 768; i += NT) slut[i] = lut[i];
    if (nout > 0) {
        if (vpass) {  // this band's 16 windows and coefficient rows from the crop's table
            const K1Layout L = k1_layout(c.h, c.w, c.new_h, c.new_w);
            const Taps* vt = (const Taps*)(tab + c.tab_off + L.vt_off);
            const int* vk = (const int*)(tab + c.tab_off + L.vk_off);
            if (tid < nout) taps[tid] = vt[y_first + tid];
            for (int i = tid; i < nout * L.kv; i += NT) {
                const int ky = i / L.kv, x = i - ky * L.kv;
                kk[ky * kvs + x] = vk[(int64_t)(y_first + ky) * L.kv + x];
            }
        } else {  // no vertical pass: a one-tap "filter" with weight 1.0 copies exactly ((p << 22) + (1 << 21)) >> 22 == p
            if (tid < nout) {
                taps[tid] = Taps{y_first + tid, 1};
                kk[tid * kvs] = 1 << PRECISION_BITS;
            }
        }
    }
    dma_wait_all();
    __syncthreads();
    if (form_a) {  // scratch rows of 672 bytes are contiguous too
        for (int e = tid; e < VIT_PATCH * ROW / 16; e += NT) ((uint4*)canvas)[e] = ((const uint4*)band)[e];
    } else if (form_b) {
        const int ncol4 = (src_row_bytes + 3) >> 2;
        uint32_t acc[NIT][4];
#pragma unroll
        for (int i = 0; i < NIT; ++i) acc[i][0] = acc[i][1] = acc[i][2] = acc[i][3] = 1u << (PRECISION_BITS - 1);
        const uint32_t* win = (const uint32_t*)window;
        for (int c0 = r0; c0 < r1; c0 += rows_chunk) {
            const int c1 = min(c0 + rows_chunk, r1);
            if (c0 != r0) {
                __syncthreads();  // every read of the previous chunk is done
                dma_range_to_lds<NT>((const uint4*)(src + (int64_t)c0 * pitch), (char*)window, (c1 - c0) * (pitch >> 4), tid);
                dma_wait_all();
                __syncthreads();
            }
#pragma unroll
            for (int i = 0; i < NIT; ++i) {
                const int e = tid + NT * i;
                const int ky = e / ROW4, c4 = e - ky * ROW4;
                if (ky < nout && c4 < ncol4) {
                    const Taps t = taps[ky];
                    const int lo = max(t.xmin, c0), hi = min(t.xmin + t.n, c1);
                    const uint32_t* wp = win + (lo - c0) * pitch4 + c4;
                    const int* kp = kk + ky * kvs + (lo - t.xmin);
                    for (int y = 0; y < hi - lo; ++y) {
                        const uint32_t d = wp[y * pitch4];
                        const uint32_t k = (uint32_t)kp[y];
                        acc[i][0] = mad24(d & 0xff, k, acc[i][0]);
                        acc[i][1] = mad24((d >> 8) & 0xff, k, acc[i][1]);
                        acc[i][2] = mad24((d >> 16) & 0xff, k, acc[i][2]);
                        acc[i][3] = mad24(d >> 24, k, acc[i][3]);
                    }
                }
            }
        }
        // canvas: resized pixels, zero outside (pad precedes normalisation)
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            const int e = tid + NT * i;
            const int ky = e / ROW4, c4 = e - ky * ROW4;
            if (ky < VIT_PATCH) {
                uint32_t v = 0;
                if (ky < nout) {
#pragma unroll
                    for (int b = 0; b < 4; ++b)
                        if (c4 * 4 + b < src_row_bytes) v |= clip8u(acc[i][b]) << (8 * b);
                }
                ((uint32_t*)canvas)[e] = v;
            }
        }
    } else {
        // width not resized: rows of the packed crop (any alignment), bytes straight from global; also the all-padding band
        for (int e = tid; e < VIT_PATCH * ROW; e += NT) {
            const int ky = e / ROW, rem = e - ky * ROW;
            uint8_t v = 0;
            if (ky < nout && rem < src_row_bytes) {
                const Taps t = taps[ky];
                const int* k = kk + ky * kvs;
                const int64_t spitch = hpass ? pitch : src_row_bytes;
                const uint8_t* p = src + (int64_t)t.xmin * spitch + rem;
                uint32_t ss0 = 1u << (PRECISION_BITS - 1);
                for (int y = 0; y < t.n; ++y) ss0 = mad24(p[(int64_t)y * spitch], (uint32_t)k[y], ss0);
                v = (uint8_t)clip8u(ss0);
            }
            canvas[e] = v;
        }
    }
    __syncthreads();
    // 14 patches x 768 values (im2col order (c, ky, kx))
    bf16_t* out = patches + ((int64_t)crop * VIT_NP + py * VIT_GRID) * VIT_D;
    if (affine) {
        // A thread emits 8 consecutive kx of one (patch, ky) for ALL three channels: 24 contiguous canvas bytes (8-byte
        // aligned: (16 px + 8 half) * 3) as three 8-byte LDS reads, every byte converted in place (v_cvt_f32_ubyteN) and
        // normalised by one fma -- verified bit-exact against the table after the bf16 rounding (NormAffine) -- then three
        // 16-byte stores, one per channel plane of the patch row.  (The table form below reads the canvas byte by byte
        // and the table at 64 data-dependent addresses: 47 % of the LDS cycles were bank conflicts.)
        for (int e = tid; e < VIT_GRID * VIT_PATCH * 2; e += NT) {
            const int px = e >> 5, ky = (e >> 1) & 15, kx0 = (e & 1) * 8;
            const uint2* cp = (const uint2*)(canvas + ky * ROW + (px * VIT_PATCH + kx0) * 3);
            const uint2 w0 = cp[0], w1 = cp[1], w2 = cp[2];
            const uint32_t w[6] = {w0.x, w0.y, w1.x, w1.y, w2.x, w2.y};
            bf16x8 o[3];
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int ch = 0; ch < 3; ++ch) {
                    const int byte = j * 3 + ch;
                    const float v = (float)((w[byte >> 2] >> (8 * (byte & 3))) & 0xffu);
                    o[ch][j] = (bf16_t)fmaf(v, aff.a[ch], aff.b[ch]);
                }
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) *(bf16x8*)(out + (int64_t)px * VIT_D + ch * 256 + ky * 16 + kx0) = o[ch];
        }
        return;
    }
    // table form: a thread emits 8 consecutive kx of one (patch, c, ky)
    for (int e = tid; e < VIT_GRID * VIT_D / 8; e += NT) {
        const int px = e / (VIT_D / 8), q = e - px * (VIT_D / 8);
        const int ch = q >> 5, ky = (q >> 1) & 15, kx0 = (q & 1) * 8;
        const uint8_t* cp = canvas + ky * ROW + (px * VIT_PATCH + kx0) * 3 + ch;
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (bf16_t)slut[ch * 256 + cp[j * 3]];
        *(bf16x8*)(out + (int64_t)px * VIT_D + q * 8) = o;
    }
}

// Mllama multi-tile output: vertical pass + zero pad to the tile canvas + normalise +
// split into tiles, f32 channel-planar [n, max_tiles, 3, T, T] as transformers'
// image_processing_pil_mllama.py:505-517 produces it (pad BEFORE normalise, tiles row-major over the
// canvas, unused tile slots all zero).  One workgroup = 8 rows of one tile slot.
constexpr int TILE_ROWS = 8;
__global__ __launch_bounds__(256) void resize_v_tiles(const uint8_t* __restrict__ pix, const uint8_t* __restrict__ tmp,
                                                      const CropDesc* __restrict__ crops, const int2* __restrict__ grid_of,
                                                      const float* __restrict__ lut, float* __restrict__ out, int T, int max_tiles) {
    __shared__ Taps taps[TILE_ROWS];
    __shared__ int kk[TILE_ROWS * MAX_TAPS];
    __shared__ float slut[3 * 256];
    const int blocks_per_tile = T / TILE_ROWS;
    const int rb = blockIdx.x % blocks_per_tile;
    const int slot = (blockIdx.x / blocks_per_tile) % max_tiles;
    const int crop = blockIdx.x / (blocks_per_tile * max_tiles);
    const CropDesc c = crops[crop];
    const int th = grid_of[crop].x, tw = grid_of[crop].y;
    const int tid = threadIdx.x;
    float* o = out + (((int64_t)crop * max_tiles + slot) * 3) * T * T + (int64_t)rb * TILE_ROWS * T;
    if (slot >= th * tw) {  // image_processing_pil_mllama.py:117-131: the tile axis is zero padded
        for (int e = tid; e < 3 * TILE_ROWS * T; e += 256) {
            const int ch = e / (TILE_ROWS * T), rem = e - ch * (TILE_ROWS * T);
            o[(int64_t)ch * T * T + rem] = 0.f;
        }
        return;
    }
    for (int i = tid; i < 768; i += 256) slut[i] = lut[i];
    const int ty = slot / tw, tx = slot - ty * tw;
    const int y0 = ty * T + rb * TILE_ROWS, x0 = tx * T;
    const bool hpass = c.new_w != c.w, vpass = c.new_h != c.h;
    const uint8_t* src = hpass ? tmp + c.tmp_off : pix + c.src_off;
    const int64_t srb = hpass ? (int64_t)k1_tmp_pitch(c.new_w) : (int64_t)c.new_w * 3;  // scratch rows are 16-byte aligned
    if (vpass && tid < TILE_ROWS && y0 + tid < c.new_h) taps[tid] = compute_taps(c.h, c.new_h, y0 + tid, kk + tid * MAX_TAPS);
    __syncthreads();
    for (int e = tid; e < 3 * TILE_ROWS * T; e += 256) {
        const int ch = e / (TILE_ROWS * T), rem = e - ch * (TILE_ROWS * T);
        const int r = rem / T, x = rem - r * T;
        const int yy = y0 + r, xx = x0 + x;
        int v = 0;
        if (yy < c.new_h && xx < c.new_w) {
            const uint8_t* p = src + (int64_t)xx * 3 + ch;
            if (!vpass) {
                v = p[yy * srb];
            } else {
                const Taps t = taps[r];
                const int* k = kk + r * MAX_TAPS;
                uint32_t ss0 = 1u << (PRECISION_BITS - 1);
                p += t.xmin * srb;
                for (int y = 0; y < t.n; ++y) ss0 = mad24(p[y * srb], (uint32_t)k[y], ss0);
                v = (int)clip8u(ss0);
            }
        }
        o[(int64_t)ch * T * T + rem] = slut[ch * 256 + v];
    }
}

// K0: cut every bounding box of ONE decoded page into the packed crop buffer K1 reads.
// Restates DocLayoutDetector.get_region_image (doclayout_detector.py:178-189): the box corners
// are already int()-truncated by the caller; `image.crop` keeps the box size and fills what
// lies outside the page with zeros.  One work item = a run of rows of one box (~32 KiB), a
// plain byte gather: HBM bound, 2 x box bytes.
__global__ __launch_bounds__(256) void crop_boxes(const uint8_t* __restrict__ page, int H, int W, const int32_t* __restrict__ boxes,
                                                  const int64_t* __restrict__ offs, const HWork* __restrict__ work,
                                                  uint8_t* __restrict__ pix) {
    const HWork wk = work[blockIdx.x];
    const int x0 = boxes[4 * wk.crop], y0 = boxes[4 * wk.crop + 1], x1 = boxes[4 * wk.crop + 2];
    const int row_bytes = (x1 - x0) * 3;
    uint8_t* dst = pix + offs[wk.crop] + (int64_t)wk.row0 * row_bytes;
    const int total = wk.nrows * row_bytes;
    for (int e = threadIdx.x; e < total; e += 256) {
        const int r = e / row_bytes, b = e - r * row_bytes;
        const int y = y0 + wk.row0 + r;
        const int xb = x0 * 3 + b;  // byte column inside the page row
        uint8_t v = 0;
        if (y >= 0 && y < H && xb >= 0 && xb < W * 3) v = page[((int64_t)y * W) * 3 + xb];
        dst[e] = v;
    }
}

}  // namespace

hipError_t launch_crop_boxes(const uint8_t* page, int H, int W, const int32_t* boxes, const int64_t* offs, const HWork* work, int nwork,
                             uint8_t* pix, hipStream_t s) {
    if (nwork <= 0) return hipSuccess;
    hipLaunchKernelGGL(crop_boxes, dim3(nwork), dim3(256), 0, s, page, H, W, boxes, offs, work, pix);
    return hipGetLastError();
}

hipError_t launch_resize_v_tiles(const uint8_t* pix, const uint8_t* tmp, const CropDesc* crops, const int32_t* grid_of, int n,
                                 const float* lut, float* out, int T, int max_tiles, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    if (T % TILE_ROWS != 0 || max_tiles < 1) return hipErrorInvalidValue;
    hipLaunchKernelGGL(resize_v_tiles, dim3((unsigned)((int64_t)n * max_tiles * (T / TILE_ROWS))), dim3(256), 0, s, pix, tmp, crops,
                       (const int2*)grid_of, lut, out, T, max_tiles);
    return hipGetLastError();
}

hipError_t launch_resample_tables(const CropDesc* crops, int n, uint8_t* tab, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(resample_tables, dim3(n), dim3(256), 0, s, crops, tab);
    return hipGetLastError();
}

hipError_t launch_resize_h(const uint8_t* pix, uint8_t* tmp, const CropDesc* crops, const HWork* work, int nwork, int lds_bytes,
                           int cls, const uint8_t* tab, hipStream_t s) {
    if (nwork <= 0) return hipSuccess;
    // lds_bytes = (padded table +) one band; + alignment lead (<= 15) + vector rounding (<= 15) + the last tap group's
    // over-read (<= 9 bytes, zero weights) + the DMA sweep's slack
    const size_t smem = (size_t)lds_bytes + 64 + DMA_SLACK;
    if (smem > 160 * 1024 || cls < 0 || cls > 2) return hipErrorInvalidValue;
    const void* fn = cls == 0 ? (const void*)resize_h<true, K1_H_RPT> : (cls == 1 ? (const void*)resize_h<true, K1_H_RPT_WIDE> : (const void*)resize_h<false, K1_H_RPT_WIDE>);
    if (hipError_t e = ensure_dynamic_lds(fn, (int)smem); e != hipSuccess) return e;
    if (cls == 0)
        hipLaunchKernelGGL((resize_h<true, K1_H_RPT>), dim3(nwork), dim3(256), smem, s, pix, tmp, crops, work, tab);
    else if (cls == 1)
        hipLaunchKernelGGL((resize_h<true, K1_H_RPT_WIDE>), dim3(nwork), dim3(256), smem, s, pix, tmp, crops, work, tab);
    else
        hipLaunchKernelGGL((resize_h<false, K1_H_RPT_WIDE>), dim3(nwork), dim3(256), smem, s, pix, tmp, crops, work, tab);
    return hipGetLastError();
}

hipError_t launch_resize_v_patchify(const uint8_t* pix, const uint8_t* tmp, const CropDesc* crops, int n, const float* lut, const NormAffine& aff,
                                    void* patches, bool any_resize, const uint8_t* tab, int kv_max, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    // the LDS window is only needed when some crop is resized or partially fills the canvas; the
    // all-224x224 batch keeps the small footprint (more workgroups per CU for a pure stream)
    static const int win_kb = diag_env("MME_K1_VWIN") ? atoi(diag_env("MME_K1_VWIN")) : 16;  // tuning switch (KiB per chunk)
    const int kvs = kv_max < 4 ? 4 : ((kv_max + 3) & ~3);
    if (kvs > MAX_TAPS) return hipErrorInvalidValue;
    const int kk_bytes = VIT_PATCH * kvs * (int)sizeof(int);
    if (any_resize) {
        const int window = (win_kb < 1 ? 1 : (win_kb > 96 ? 96 : win_kb)) * 1024;
        const int smem = kk_bytes + window + DMA_SLACK;
        if (hipError_t e = ensure_dynamic_lds((const void*)resize_v_patchify<true>, smem); e != hipSuccess) return e;
        hipLaunchKernelGGL(resize_v_patchify<true>, dim3(n * VIT_GRID), dim3(512), smem, s, pix, tmp, crops, lut, (bf16_t*)patches, tab, window, kvs, aff);
    } else {
        hipLaunchKernelGGL(resize_v_patchify<false>, dim3(n * VIT_GRID), dim3(256), kk_bytes, s, pix, tmp, crops, lut, (bf16_t*)patches, tab, 0, kvs, aff);
    }
    return hipGetLastError();
}
