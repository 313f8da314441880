// Plane-sweep sampling geometry, bit-faithful to the reference's PyTorch-CPU path.
//
// The reference builds the sampling grid from ATen ops (warping/homography.py:119-121,
// :185-196) and hands it to F.grid_sample(bilinear, zeros, align_corners=False).  A depth
// map only matches that path to 1e-4 if the sample positions match to the last bit (one
// ulp of ix at x~300 moves the L2 cost by ~1e-3), so every operation below reproduces the
// rounding of the corresponding ATen/MKL op.  The orders were pinned empirically against
// torch 2.10 CPU (tests/test_coords_pin.py keeps them pinned):
//
//   K@R, (K@R)@rays  sgemm, N>=2 : blas_mode FMA      -> fma chain over k = 0,1,2, first product
//                                                       rounded alone (MKL on Intel)
//                                  blas_mode SEPARATE -> (p0 + p1) + p2, products rounded
//                                                       separately (MKL on AMD EPYC)
//   K@t              sgemv, N==1 : FMA -> (p1 + p2) + p0 ; SEPARATE -> (p0 + p1) + p2 ; both with
//                                  separately rounded products
//   term1 + term2*d              : separate mul and add (two ATen ops)
//   P / (Pz + 1e-10)             : IEEE add, IEEE divide
//   (u - cx) / cx                : IEEE sub, IEEE divide
//   un-normalise                 : fma(g + 1, size/2, -0.5)   (vectorised CPU kernel,
//                                  ATen/native/cpu/GridSamplerKernel.cpp, contracted)
//   bilinear weights             : w = ix - floor(ix); e = 1 - w; nw = s*e, ne = s*w, ...
//   interpolation                : fma(se_v,se, fma(sw_v,sw, fma(ne_v,ne, nw_v*nw)))
//
// The translation unit is compiled with -ffp-contract=off: every fma is explicit.
#pragma once
#include <hip/hip_runtime.h>

namespace pdepth {

// Per (batch item, view) homography terms, computed once per thread from uniform loads.
struct ViewXform {
    float kr[9];  // K @ R
    float kt[3];  // K @ t
    int separate; // blas_mode == PDEPTH_BLAS_SEPARATE
};

__device__ __forceinline__ float dot3_blas(int separate, float a0, float b0, float a1, float b1,
                                           float a2, float b2) {
    const float p0 = a0 * b0;
    if (separate) {
        const float p1 = a1 * b1;
        const float p2 = a2 * b2;
        return (p0 + p1) + p2;
    }
    return __builtin_fmaf(a2, b2, __builtin_fmaf(a1, b1, p0));
}

__device__ __forceinline__ void make_view_xform(const float* __restrict__ K,
                                                const float* __restrict__ R,
                                                const float* __restrict__ t, int blas_mode,
                                                ViewXform& x) {
    x.separate = blas_mode;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
        for (int j = 0; j < 3; ++j)
            x.kr[i * 3 + j] = dot3_blas(x.separate, K[i * 3 + 0], R[0 * 3 + j], K[i * 3 + 1],
                                        R[1 * 3 + j], K[i * 3 + 2], R[2 * 3 + j]);
        const float p0 = K[i * 3 + 0] * t[0];
        const float p1 = K[i * 3 + 1] * t[1];
        const float p2 = K[i * 3 + 2] * t[2];
        x.kt[i] = x.separate ? (p0 + p1) + p2 : (p1 + p2) + p0;
    }
}

// term2 = (K@R) @ ray for one pixel.
__device__ __forceinline__ void ray_term2(const ViewXform& x, float r0, float r1, float r2,
                                          float& a, float& b, float& c) {
    a = dot3_blas(x.separate, x.kr[0], r0, x.kr[1], r1, x.kr[2], r2);
    b = dot3_blas(x.separate, x.kr[3], r0, x.kr[4], r1, x.kr[5], r2);
    c = dot3_blas(x.separate, x.kr[6], r0, x.kr[7], r1, x.kr[8], r2);
}

// Un-normalised sample position of one pixel on one depth plane.
__device__ __forceinline__ void plane_sample_pos(const ViewXform& x, float t2a, float t2b,
                                                 float t2c, float d, float cx, float cy,
                                                 float half_w, float half_h, float& ix,
                                                 float& iy) {
    const float px = x.kt[0] + t2a * d;
    const float py = x.kt[1] + t2b * d;
    const float pz = x.kt[2] + t2c * d;
    const float den = pz + 1e-10f;
    const float u = px / den;
    const float v = py / den;
    const float gx = (u - cx) / cx;
    const float gy = (v - cy) / cy;
    ix = __builtin_fmaf(gx + 1.0f, half_w, -0.5f);
    iy = __builtin_fmaf(gy + 1.0f, half_h, -0.5f);
}

// The same position with the divides spelled out.  hipcc lowers an IEEE fp32 divide n/d to
//     d' = div_scale(d), n' = div_scale(n), y0 = rcp(d'), e = fma(-d', y0, 1), y = fma(e, y0, y0),
//     q0 = n'*y, r0 = fma(-d', q0, n'), q1 = fma(r0, y, q0), r1 = fma(-d', q1, n'),
//     q = div_fmas(r1, y, q1), div_fixup
// where div_scale / div_fmas / div_fixup only act on operands near the ends of the exponent range (they
// are the identity for |d|, |n/d| within 2^+-96 or so and finite).  Positions in that range -- every
// position that can land within reach of the image -- therefore come out bit-identical from the plain
// fma chain below, and it lets the two divides by den share y (rcp + 2 fma) and the divides by the
// wave-uniform cx, cy use reciprocals refined once per thread (refined_rcp).  Outside the range the
// chain yields NaN or a huge value where IEEE yields inf / a huge value; both classify as "all taps
// out of bounds" downstream, with NaN weights in exactly the cases where the reference has them
// (inf - floor(inf) = NaN).  tests/test_hip_parity.py compares both variants bit for bit.
__device__ __forceinline__ float refined_rcp(float d) {
    const float y0 = __builtin_amdgcn_rcpf(d);
    const float e = __builtin_fmaf(-d, y0, 1.0f);
    return __builtin_fmaf(e, y0, y0);
}
__device__ __forceinline__ float div_core(float n, float d, float y) {
    const float q0 = n * y;
    const float r0 = __builtin_fmaf(-d, q0, n);
    const float q1 = __builtin_fmaf(r0, y, q0);
    const float r1 = __builtin_fmaf(-d, q1, n);
    return __builtin_fmaf(r1, y, q1);
}
__device__ __forceinline__ void plane_sample_pos_fast(const ViewXform& x, float t2a, float t2b,
                                                      float t2c, float d, float cx, float cy,
                                                      float rcx, float rcy, float half_w,
                                                      float half_h, float& ix, float& iy) {
    const float px = x.kt[0] + t2a * d;
    const float py = x.kt[1] + t2b * d;
    const float pz = x.kt[2] + t2c * d;
    const float den = pz + 1e-10f;
    const float y = refined_rcp(den);
    const float u = div_core(px, den, y);
    const float v = div_core(py, den, y);
    const float gx = div_core(u - cx, cx, rcx);
    const float gy = div_core(v - cy, cy, rcy);
    ix = __builtin_fmaf(gx + 1.0f, half_w, -0.5f);
    iy = __builtin_fmaf(gy + 1.0f, half_h, -0.5f);
}

// exp(x) for x <= 0 (softmax: x = cost - max): 2^(x log2 e) with the product split into a rounded head and its
// exact remainder, so that the argument of v_exp_f32 is exact to an ulp for every x the softmax can produce.  ~1.5 ulp
// (libm's expf: 1 ulp) in 9 instructions instead of 20.  NaN stays NaN, anything below -1000 gives 0.
__device__ __forceinline__ float exp_nonpos(float x) {
    x = x < -1000.0f ? -1000.0f : x;   // (keeps -inf out of the fma below)
    const float L2E_HI = 1.44269502162933349609375f, L2E_LO = 1.925963033500011079e-8f;
    const float t = x * L2E_HI;
    const float tl = __builtin_fmaf(x, L2E_HI, -t) + x * L2E_LO;   // x log2 e = t + tl
    const float ti = rintf(t);
    const float p = __builtin_amdgcn_exp2f((t - ti) + tl);           // argument in [-0.5, 0.5]
    return ldexpf(p, (int)ti);
}

// Bilinear footprint of a sample position: top-left texel, the four weights and a 4-bit
// in-bounds mask (bit0 nw, bit1 ne, bit2 sw, bit3 se).  Positions that are NaN or far
// outside the image get mask 0 (all taps read as zero, like padding_mode='zeros').
struct Footprint {
    int x0, y0;
    float nw, ne, sw, se;
    unsigned mask;
};

__device__ __forceinline__ Footprint make_footprint(float ix, float iy, int W, int H) {
    Footprint f;
    const float xf = floorf(ix);
    const float yf = floorf(iy);
    const float w = ix - xf;
    const float e = 1.0f - w;
    const float n = iy - yf;
    const float s = 1.0f - n;
    f.nw = s * e;
    f.ne = s * w;
    f.sw = n * e;
    f.se = n * w;
    // Clamp before the int conversion so huge / NaN positions cannot overflow; anything at
    // or beyond -2 / size+1 is fully out of bounds anyway.
    const float xc = fminf(fmaxf(xf, -2.0f), (float)(W + 1));
    const float yc = fminf(fmaxf(yf, -2.0f), (float)(H + 1));
    const bool finite = (ix == ix) && (iy == iy);
    f.x0 = (int)xc;
    f.y0 = (int)yc;
    const bool xin0 = f.x0 >= 0 && f.x0 < W;
    const bool xin1 = f.x0 + 1 >= 0 && f.x0 + 1 < W;
    const bool yin0 = f.y0 >= 0 && f.y0 < H;
    const bool yin1 = f.y0 + 1 >= 0 && f.y0 + 1 < H;
    unsigned m = 0;
    m |= (xin0 && yin0) ? 1u : 0u;
    m |= (xin1 && yin0) ? 2u : 0u;
    m |= (xin0 && yin1) ? 4u : 0u;
    m |= (xin1 && yin1) ? 8u : 0u;
    f.mask = finite ? m : 0u;
    return f;
}

}  // namespace pdepth
