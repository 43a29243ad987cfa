// Fused GEMM epilogues shared by the 128x128 and 256x256 MFMA kernels.
//
// A lane always owns 4 consecutive output columns n..n+3 of one row m (operands are fed to
// the MFMA swapped so that the accumulator's register index runs along n): bias / residual /
// position vectors load as one 16- or 8-byte vector and the result stores as bf16x4 (8 B) or
// f32x4 (16 B).
#pragma once
#include "common.h"
#include "kernels.h"

// erf-GELU (transformers ACT2FN["gelu"], modeling_vit.py:241-255):
//   gelu(x) = x * Phi(x) = relu(x) - |x| * q(|x|),   q(a) = 0.5 * erfc(a / sqrt(2)).
// erfc by Abramowitz-Stegun 7.1.26 (|err| < 1.5e-7): erfc(z) = t*P4(t)*exp(-z^2), t = 1/(1+p*z).
// Written for the fewest VALU issues (the fc1 epilogue evaluates 32768 of these per wave-tile
// and is VALU bound): 11 plain operations that the compiler pairs into packed f32 math, plus
// one rcp and one exp2; no branches, no sign fix-up.  The result is rounded to bf16 right after.
__device__ __forceinline__ float gelu_erf(float x) {
    const float a = fabsf(x);
    const float t = __frcp_rn(fmaf(a, 0.3275911f * 0.70710678118654752f, 1.0f));
    float h = fmaf(t, 0.5f * 1.061405429f, 0.5f * -1.453152027f);
    h = fmaf(h, t, 0.5f * 1.421413741f);
    h = fmaf(h, t, 0.5f * -0.284496736f);
    h = fmaf(h, t, 0.5f * 0.254829592f);
    h *= t;
    const float e = __builtin_amdgcn_exp2f((x * -0.72134752044448170f) * x);  // exp(-x^2/2)
    return fmaf(-a, h * e, fmaxf(x, 0.0f));
}

struct EpiRow {
    int64_t orow;  // output row (EPI_PATCH remaps patch rows past the [CLS] rows)
    int prow;      // position row for EPI_PATCH
};

template <int EPI>
__device__ __forceinline__ EpiRow epi_row(int m) {
    EpiRow r{m, 0};
    if (EPI == EPI_PATCH) {
        const int b = m / VIT_NP;
        r.prow = m - b * VIT_NP + 1;
        r.orow = (int64_t)b * VIT_T + r.prow;
    }
    return r;
}

// v = accumulator values for (m, n..n+3); n < N and n % 4 == 0 guaranteed by the caller.
template <int EPI>
__device__ __forceinline__ void epi_store(const GemmArgs& g, int m, const EpiRow& er, int n, f32x4 v) {
    if (EPI == EPI_F32) {
        float* o = g.outf + (int64_t)m * g.ldf + n;
        if (n + 3 < g.N && ((g.ldf & 3) == 0)) {
            *(f32x4*)o = v;
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (n + r < g.N) o[r] = v[r];
        }
        return;
    }
    v += *(const f32x4*)(g.bias + n);
    if (EPI == EPI_BIAS_GELU) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = gelu_erf(v[r]);
    }
    if (EPI == EPI_PATCH) v += *(const f32x4*)(g.pos + (int64_t)er.prow * g.N + n);
    bf16_t* o = (bf16_t*)g.out + er.orow * g.ldo + n;
    if (EPI == EPI_BIAS_RES) {
        const bf16x4 rv = *(const bf16x4*)((const bf16_t*)g.res + er.orow * g.ldo + n);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] += (float)rv[r];
    }
    bf16x4 ov;
#pragma unroll
    for (int r = 0; r < 4; ++r) ov[r] = (bf16_t)v[r];
    *(bf16x4*)o = ov;
}

// ---- 16-byte epilogue accesses (halves the number of store / load instructions) -------------
// The epilogue of a 256x256 tile is store-ISSUE bound (one CU retires a vector store
// instruction every few tens of cycles whatever its width), so width is what counts.  A lane
// owns 4 consecutive columns (8 bytes of bf16) of a 16x16 accumulator tile; lanes fq and fq^1
// (16 lanes apart) own the neighbouring 4.  v_permlane16_swap exchanges the odd 16-lane rows of
// one register with the even rows of another: applied to the packed data of two adjacent
// column tiles (j, j+1) it leaves every lane with 8 consecutive columns of ONE tile:
//   fq even -> tile j,   columns 8*(fq>>1) .. +7      fq odd -> tile j+1, same columns
// The same swap applied to 16 bytes loaded in that layout returns them to accumulator layout.
__device__ __forceinline__ uint4 pair_to_row16(uint2 p, uint2 q) {
    const auto a = __builtin_amdgcn_permlane16_swap(p.x, q.x, false, false);
    const auto b = __builtin_amdgcn_permlane16_swap(p.y, q.y, false, false);
    return make_uint4(a[0], b[0], a[1], b[1]);
}
__device__ __forceinline__ void row16_to_pair(uint4 x, uint2& p, uint2& q) {
    const auto a = __builtin_amdgcn_permlane16_swap(x.x, x.z, false, false);
    const auto b = __builtin_amdgcn_permlane16_swap(x.y, x.w, false, false);
    p = make_uint2(a[0], b[0]);
    q = make_uint2(a[1], b[1]);
}
__device__ __forceinline__ uint2 pack_bf16x4(f32x4 v) {
    bf16x4 o;
#pragma unroll
    for (int r = 0; r < 4; ++r) o[r] = (bf16_t)v[r];
    return __builtin_bit_cast(uint2, o);
}
__device__ __forceinline__ f32x4 unpack_bf16x4(uint2 u) {
    const bf16x4 b = __builtin_bit_cast(bf16x4, u);
    return f32x4{(float)b[0], (float)b[1], (float)b[2], (float)b[3]};
}
// column (within the wave's 64) of this lane's 16-byte piece for the tile pair (jp, jp+1)
__device__ __forceinline__ int row16_col(int jp, int fq) { return ((fq & 1) ? (jp + 1) * 16 : jp * 16) + (fq >> 1) * 8; }
