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
//   resize_h          one workgroup per (crop, band of source rows): source row -> LDS,
//                     fixed-point taps from an LDS coefficient table -> uint8 scratch row.
//   resize_v_patchify one workgroup per (crop, patch row): vertical taps (or a plain copy
//                     when the height is unchanged, e.g. the 224x224 synthetic crops) into a
//                     16 x 224 x 3 uint8 LDS canvas, then LUT-normalise and emit 14 patches.
//
// Coefficients are computed on the device in f64 exactly as Resample.c does on the host;
// contraction is disabled so no fused multiply-add changes a rounding.
#include "common.h"
#include "kernels.h"

#pragma clang fp contract(off)

namespace {

constexpr int PRECISION_BITS = 32 - 8 - 2;
constexpr int MAX_TAPS = 160;      // window <= 2*ceil(scale)+1 and scale < 2*MAX_DIM/224
constexpr int V_WINDOW = 36 * 1024; // bytes of source rows a resize_v workgroup stages in LDS

struct Taps {
    int xmin, n;
};

// One output coordinate's window and fixed-point weights (Resample.c precompute_coeffs +
// normalize_coeffs_8bpc, bilinear filter, box = whole image).
__device__ __forceinline__ Taps compute_taps(int in_size, int out_size, int xx, int* kk /*[MAX_TAPS]*/) {
    const double scale = (double)in_size / (double)out_size;
    const double filterscale = scale < 1.0 ? 1.0 : scale;
    const double support = 1.0 * filterscale;
    const double ss = 1.0 / filterscale;
    const double center = (xx + 0.5) * scale;
    int xmin = (int)(center - support + 0.5);
    if (xmin < 0) xmin = 0;
    int xmax = (int)(center + support + 0.5);
    if (xmax > in_size) xmax = in_size;
    const int n = xmax - xmin;
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
        kk[x] = w < 0 ? (int)(-0.5 + w * (double)(1 << PRECISION_BITS)) : (int)(0.5 + w * (double)(1 << PRECISION_BITS));
    }
    return Taps{xmin, n};
}

__device__ __forceinline__ uint8_t clip8(int v) {
    v >>= PRECISION_BITS;
    return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// Tap tables of the horizontal pass, ONCE per crop.  A band workgroup used to recompute its crop's table (f64 loops over
// up to 2 * scale + 1 taps per output column: ~3 k cycles, more than the filtering of a 2-row band of a wide page region
// costs); now it copies the table from L2 with 16-byte loads.
__global__ __launch_bounds__(256) void h_tables(const CropDesc* __restrict__ crops, uint8_t* __restrict__ tab) {
    const CropDesc c = crops[blockIdx.x];
    if (c.new_w == c.w) return;
    const int kstride = 2 * ((c.w + c.new_w - 1) / c.new_w) + 1;
    Taps* taps = (Taps*)(tab + c.tab_off);
    int* kk = (int*)(taps + c.new_w);
    for (int x = threadIdx.x; x < c.new_w; x += 256) taps[x] = compute_taps(c.w, c.new_w, x, kk + x * kstride);
}

// Horizontal pass.  One workgroup = one band of source rows of one crop; the band is a single
// contiguous byte range (rows are contiguous), fetched with one sweep of 16-byte loads into LDS
// (all loads in flight at once), then every (row, x, channel) output of the band is computed
// from LDS in parallel.  Band height is chosen on the host so that a band is <= H_BAND bytes.
__global__ __launch_bounds__(256) void resize_h(const uint8_t* __restrict__ pix, uint8_t* __restrict__ tmp,
                                                const CropDesc* __restrict__ crops, const HWork* __restrict__ work, int table_ints, int taps_cap,
                                                const uint8_t* __restrict__ tab) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const HWork wk = work[blockIdx.x];
    const CropDesc c = crops[wk.crop];
    const int tid = threadIdx.x;
    const int row_bytes = c.w * 3;
    // LDS carve: [taps_cap] Taps (>= widest output row of the batch) | coefficient table (per-launch size) | band
    Taps* taps = (Taps*)smem;
    int* kk = (int*)(smem + (size_t)taps_cap * sizeof(Taps));
    uint8_t* band = (uint8_t*)(smem + (size_t)taps_cap * sizeof(Taps) + (size_t)table_ints * sizeof(int));
    const int kstride = 2 * ((c.w + c.new_w - 1) / c.new_w) + 1;  // >= 2*ceil(max(scale,1))+1
    {   // the crop's precomputed table (h_tables): {xmin, n} pairs, then the coefficients -- two 16-byte-aligned sweeps
        const uint4* gt = (const uint4*)(tab + c.tab_off);
        const int ntap_vec = (c.new_w * (int)sizeof(Taps) + 15) >> 4;
        for (int i = tid; i < ntap_vec; i += 256) ((uint4*)taps)[i] = gt[i];
        // (the coefficient block starts 8 * new_w bytes in: 16-byte aligned only for even new_w -> copy as 8-byte words)
        const uint2* gk = (const uint2*)(tab + c.tab_off + (size_t)c.new_w * sizeof(Taps));
        for (int i = tid; i < (c.new_w * kstride * 4 + 7) >> 3; i += 256) ((uint2*)kk)[i] = gk[i];
    }
    const uint8_t* src = pix + c.src_off + (int64_t)wk.row0 * row_bytes;
    const int nbytes = wk.nrows * row_bytes;
    const uintptr_t a0 = (uintptr_t)src & ~(uintptr_t)15;
    const int lead = (int)((uintptr_t)src - a0);
    const int nvec = (lead + nbytes + 15) >> 4;
    for (int i = tid; i < nvec; i += 256) ((uint4*)band)[i] = ((const uint4*)a0)[i];
    __syncthreads();
    const uint8_t* bb = band + lead;
    const int out_row = c.new_w * 3;
    uint8_t* dst = tmp + c.tmp_off + (int64_t)wk.row0 * out_row;
    // one output PIXEL COLUMN per thread for a PAIR of rows (y, y + half): the six sums share every coefficient read and give
    // the LDS byte reads of one row something independent to overlap with
    const int half = (wk.nrows + 1) >> 1;
    for (int e = tid; e < half * c.new_w; e += 256) {
        const int y = e / c.new_w, xx = e - y * c.new_w;
        const int y2 = y + half;
        const bool two = y2 < wk.nrows;
        const Taps t = taps[xx];
        const int* k = kk + xx * kstride;
        int s0 = 1 << (PRECISION_BITS - 1), s1 = s0, s2 = s0, u0 = s0, u1 = s0, u2 = s0;
        const uint8_t* p = bb + y * row_bytes + t.xmin * 3;
        const uint8_t* p2 = two ? p + half * row_bytes : p;
        for (int x = 0; x < t.n; ++x) {
            const int w = k[x];
            s0 += (int)p[x * 3] * w;
            s1 += (int)p[x * 3 + 1] * w;
            s2 += (int)p[x * 3 + 2] * w;
            u0 += (int)p2[x * 3] * w;
            u1 += (int)p2[x * 3 + 1] * w;
            u2 += (int)p2[x * 3 + 2] * w;
        }
        uint8_t* o = dst + ((int64_t)y * c.new_w + xx) * 3;
        o[0] = clip8(s0);
        o[1] = clip8(s1);
        o[2] = clip8(s2);
        if (two) {
            uint8_t* o2 = dst + ((int64_t)y2 * c.new_w + xx) * 3;
            o2[0] = clip8(u0);
            o2[1] = clip8(u1);
            o2[2] = clip8(u2);
        }
    }
}

__global__ __launch_bounds__(256) void resize_v_patchify(const uint8_t* __restrict__ pix, const uint8_t* __restrict__ tmp,
                                                         const CropDesc* __restrict__ crops, const float* __restrict__ lut,
                                                         bf16_t* __restrict__ patches, int window_bytes) {
    __shared__ __attribute__((aligned(16))) uint8_t canvas[VIT_PATCH * VIT_IMG * 3 + 16];
    __shared__ Taps taps[VIT_PATCH];
    __shared__ int kk[VIT_PATCH * MAX_TAPS];
    __shared__ float slut[3 * 256];
    extern __shared__ __attribute__((aligned(16))) uint8_t window[];  // source rows feeding this band (0 bytes for all-224 batches)
    const int crop = blockIdx.x / VIT_GRID, py = blockIdx.x - crop * VIT_GRID;
    const CropDesc c = crops[crop];
    const int tid = threadIdx.x;
    for (int i = tid; i < 768; i += 256) slut[i] = lut[i];
    const bool hpass = c.new_w != c.w;       // Resample.c: horizontal pass only when width changes
    const bool vpass = c.new_h != c.h;
    const uint8_t* src = hpass ? tmp + c.tmp_off : pix + c.src_off;
    const int src_row_bytes = c.new_w * 3;   // after the horizontal pass (or unchanged width)
    if (vpass && tid < VIT_PATCH) {
        const int yy = py * VIT_PATCH + tid;
        if (yy < c.new_h) taps[tid] = compute_taps(c.h, c.new_h, yy, kk + tid * MAX_TAPS);
    }
    __syncthreads();
    // fill the 16-row canvas band: resized pixels, zero outside (pad precedes normalisation)
    const uint8_t* band = src + (int64_t)py * VIT_PATCH * src_row_bytes;
    if (!vpass && src_row_bytes == VIT_IMG * 3 && (py + 1) * VIT_PATCH <= c.new_h && (((uintptr_t)band) & 15) == 0) {
        // unchanged 224-wide rows (the synthetic 224x224 workload): one contiguous 10752-byte band
        for (int e = tid; e < VIT_PATCH * VIT_IMG * 3 / 16; e += 256) ((uint4*)canvas)[e] = ((const uint4*)band)[e];
    } else
    {
        // vertical pass (or row copy): the source rows this band needs form one contiguous byte
        // range; stage it in LDS with 16-byte loads when it fits, else read global directly
        const int y_first = py * VIT_PATCH, y_last = min(y_first + VIT_PATCH, c.new_h) - 1;
        int r0 = 0, r1 = -1;  // source row range [r0, r1]
        if (y_last >= y_first) {
            if (vpass) {
                r0 = taps[0].xmin;
                r1 = taps[y_last - y_first].xmin + taps[y_last - y_first].n - 1;
            } else {
                r0 = y_first;
                r1 = y_last;
            }
        }
        const int64_t wbytes = (int64_t)(r1 - r0 + 1) * src_row_bytes;
        const uint8_t* wsrc = src + (int64_t)r0 * src_row_bytes;
        const bool staged = wbytes > 0 && wbytes <= window_bytes;
        int lead = 0;
        if (staged) {
            const uintptr_t a0 = (uintptr_t)wsrc & ~(uintptr_t)15;
            lead = (int)((uintptr_t)wsrc - a0);
            const int nvec = (int)((lead + wbytes + 15) >> 4);
            for (int i = tid; i < nvec; i += 256) ((uint4*)window)[i] = ((const uint4*)a0)[i];
        }
        __syncthreads();
        const uint8_t* wb = window + lead;
        for (int e = tid; e < VIT_PATCH * VIT_IMG * 3; e += 256) {
            const int ky = e / (VIT_IMG * 3), rem = e - ky * (VIT_IMG * 3);
            const int yy = py * VIT_PATCH + ky;
            uint8_t v = 0;
            if (yy < c.new_h && rem < src_row_bytes) {
                if (!vpass) {
                    v = staged ? wb[(yy - r0) * src_row_bytes + rem] : src[(int64_t)yy * src_row_bytes + rem];
                } else {
                    const Taps t = taps[ky];
                    const int* k = kk + ky * MAX_TAPS;
                    int ss0 = 1 << (PRECISION_BITS - 1);
                    if (staged) {
                        const uint8_t* p = wb + (t.xmin - r0) * src_row_bytes + rem;
                        for (int y = 0; y < t.n; ++y) ss0 += (int)p[y * src_row_bytes] * k[y];
                    } else {
                        const uint8_t* p = src + (int64_t)t.xmin * src_row_bytes + rem;
                        for (int y = 0; y < t.n; ++y) ss0 += (int)p[(int64_t)y * src_row_bytes] * k[y];
                    }
                    v = clip8(ss0);
                }
            }
            canvas[e] = v;
        }
    }
    __syncthreads();
    // 14 patches x 768 values; a thread emits 8 consecutive kx of one (patch, c, ky)
    bf16_t* out = patches + ((int64_t)crop * VIT_NP + py * VIT_GRID) * VIT_D;
    for (int e = tid; e < VIT_GRID * VIT_D / 8; e += 256) {
        const int px = e / (VIT_D / 8), q = e - px * (VIT_D / 8);
        const int ch = q >> 5, ky = (q >> 1) & 15, kx0 = (q & 1) * 8;
        const uint8_t* cp = canvas + ky * (VIT_IMG * 3) + (px * VIT_PATCH + kx0) * 3 + ch;
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
    const int64_t srb = (int64_t)c.new_w * 3;
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
                int ss0 = 1 << (PRECISION_BITS - 1);
                p += t.xmin * srb;
                for (int y = 0; y < t.n; ++y) ss0 += (int)p[y * srb] * k[y];
                v = clip8(ss0);
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

hipError_t launch_h_tables(const CropDesc* crops, int n, uint8_t* tab, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(h_tables, dim3(n), dim3(256), 0, s, crops, tab);
    return hipGetLastError();
}

hipError_t launch_resize_h(const uint8_t* pix, uint8_t* tmp, const CropDesc* crops, const HWork* work, int nwork, int table_ints,
                           int band_bytes, const uint8_t* tab, hipStream_t s, int taps_cap) {
    if (nwork <= 0) return hipSuccess;
    // Taps table + coefficient table (largest of the batch) + one band (+ alignment slack)
    if (taps_cap < VIT_IMG) taps_cap = VIT_IMG;
    const size_t smem = (size_t)taps_cap * sizeof(Taps) + (size_t)table_ints * sizeof(int) + (size_t)band_bytes + 48;
    if (smem > 160 * 1024) return hipErrorInvalidValue;
    if (hipError_t e = ensure_dynamic_lds((const void*)resize_h, (int)smem); e != hipSuccess) return e;
    hipLaunchKernelGGL(resize_h, dim3(nwork), dim3(256), smem, s, pix, tmp, crops, work, table_ints, taps_cap, tab);
    return hipGetLastError();
}

hipError_t launch_resize_v_patchify(const uint8_t* pix, const uint8_t* tmp, const CropDesc* crops, int n, const float* lut,
                                    void* patches, bool any_resize, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    // the LDS window is only needed when some crop is resized or partially fills the canvas; the
    // all-224x224 batch keeps the small footprint (more workgroups per CU for a pure stream)
    const int window = any_resize ? V_WINDOW : 0;
    hipLaunchKernelGGL(resize_v_patchify, dim3(n * VIT_GRID), dim3(256), window ? window + 32 : 0, s, pix, tmp, crops, lut,
                       (bf16_t*)patches, window);
    return hipGetLastError();
}
