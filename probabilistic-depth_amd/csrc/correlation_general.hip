// The reference's correlation operator for EVERY configuration its kernel defines (forward: models/correlation_package/
// correlation_cuda_kernel.cu:41-114; backward: the gradient of exactly that sum, which :116-300 compute over the padded
// NHWC repacks), in fp32 and in fp16 I/O with fp32 accumulation (the reference dispatches AT_DISPATCH_FLOATING_TYPES_AND_HALF,
// .cu:352-369).  The configuration the reference instantiates (kernel 1, stride1 1, pad = max_displacement: pwclite.py:123-125)
// has its own LDS-tiled fp32 kernels in extras.hip; this file is the general path.
//
// Index arithmetic, as the reference's, in the coordinates of the zero-padded inputs (pad_size on every side):
//     kr = (kernel_size - 1) / 2,  dr = max_displacement / stride2,  ds = 2 dr + 1,  border = kr + max_displacement
//     output [B, ds*ds, oH, oW],  oH = ceil((H + 2 pad - 2 border) / stride1)
//     (y1, x1) = (oy, ox) * stride1 + max_displacement          -- NOT + border: the reference's own offset
//     out[b, (tj+dr)*ds + (ti+dr), oy, ox] = 1 / (k*k*C) * sum_{j,i in [-kr,kr]} sum_c
//                                            in1p[b, c, y1+j, x1+i] * in2p[b, c, y1 + tj*stride2 + j, x1 + ti*stride2 + i]
// No repack: a padded coordinate p maps to the image coordinate p - pad, zero outside.  Configurations whose indices
// would leave the padded buffer (the reference reads out of bounds there) are refused by the C ABI.
//
//   forward : thread = one output element, lanes along ox (coalesced reads when stride1 = 1), channels in a loop;
//   backward: thread = one input element, a GATHER over the outputs it contributed to (no atomics, deterministic):
//       d in1[c,y,x] = 1/n sum_{j,i} [oy, ox integral and in range] sum_{tj,ti} go[tc, oy, ox] * in2[c, y + tj*s2, x + ti*s2]
//       d in2[c,y,x] = 1/n sum_{tj,ti} sum_{j,i} [..]                   go[tc, oy, ox] * in1[c, y - tj*s2, x - ti*s2]
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>

#include "kernels.hpp"

namespace pdepth {

namespace {

struct CorrCfg {
    int B, C, H, W, pad, kr, dr, ds, md, s1, s2, oH, oW;
    float inv_n;
};

template <typename T> __device__ __forceinline__ float ldf(const T* p, long long i) { return (float)p[i]; }
template <> __device__ __forceinline__ float ldf<__half>(const __half* p, long long i) { return __half2float(p[i]); }
template <typename T> __device__ __forceinline__ void stf(T* p, long long i, float v) { p[i] = (T)v; }
template <> __device__ __forceinline__ void stf<__half>(__half* p, long long i, float v) { p[i] = __float2half(v); }

template <typename T>
__global__ __launch_bounds__(256) void corr_general_fwd(const T* __restrict__ in1, const T* __restrict__ in2, T* __restrict__ out, CorrCfg g) {
    const long long total = (long long)g.B * g.ds * g.ds * g.oH * g.oW;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int ox = (int)(idx % g.oW), oy = (int)((idx / g.oW) % g.oH);
    const int tc = (int)((idx / ((long long)g.oW * g.oH)) % (g.ds * g.ds)), b = (int)(idx / ((long long)g.oW * g.oH * g.ds * g.ds));
    const int tj = tc / g.ds - g.dr, ti = tc % g.ds - g.dr;
    const int y1 = oy * g.s1 + g.md - g.pad, x1 = ox * g.s1 + g.md - g.pad;   // image coordinates of the window centres
    const int y2 = y1 + tj * g.s2, x2 = x1 + ti * g.s2;
    const long long HW = (long long)g.H * g.W;
    const T* a = in1 + (long long)b * g.C * HW;
    const T* c2 = in2 + (long long)b * g.C * HW;
    float acc = 0.0f;
    for (int j = -g.kr; j <= g.kr; ++j)
        for (int i = -g.kr; i <= g.kr; ++i) {
            const int ya = y1 + j, xa = x1 + i, yb = y2 + j, xb = x2 + i;
            if ((unsigned)ya >= (unsigned)g.H || (unsigned)xa >= (unsigned)g.W || (unsigned)yb >= (unsigned)g.H || (unsigned)xb >= (unsigned)g.W)
                continue;   // a zero of the padding on either side
            const long long pa = (long long)ya * g.W + xa, pb = (long long)yb * g.W + xb;
            for (int c = 0; c < g.C; ++c) acc = __builtin_fmaf(ldf(a, c * HW + pa), ldf(c2, c * HW + pb), acc);
        }
    stf(out, idx, acc * g.inv_n);
}

// WHICH = 1: gradient of input1 (gathers input2), WHICH = 2: gradient of input2 (gathers input1)
template <typename T, int WHICH>
__global__ __launch_bounds__(256) void corr_general_bwd(const T* __restrict__ other, const T* __restrict__ go, T* __restrict__ grad, CorrCfg g) {
    const long long HW = (long long)g.H * g.W, total = (long long)g.B * g.C * HW;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int x = (int)(idx % g.W), y = (int)((idx / g.W) % g.H), c = (int)((idx / HW) % g.C), b = (int)(idx / (HW * g.C));
    const T* o = other + ((long long)b * g.C + c) * HW;
    const T* gb = go + (long long)b * g.ds * g.ds * g.oH * g.oW;
    float acc = 0.0f;
    for (int tj = -g.dr; tj <= g.dr; ++tj)
        for (int ti = -g.dr; ti <= g.dr; ++ti) {
            // the partner texel in the other input, and the window centre (y1, x1) in padded coordinates of an output that
            // pairs this element with it through kernel offset (j, i)
            const int yo = WHICH == 1 ? y + tj * g.s2 : y - tj * g.s2, xo = WHICH == 1 ? x + ti * g.s2 : x - ti * g.s2;
            if ((unsigned)yo >= (unsigned)g.H || (unsigned)xo >= (unsigned)g.W) continue;
            const float ov = ldf(o, (long long)yo * g.W + xo);
            const int tc = (tj + g.dr) * g.ds + (ti + g.dr);
            for (int j = -g.kr; j <= g.kr; ++j) {
                const int y1 = (WHICH == 1 ? y : yo) + g.pad - j - g.md;   // = oy * stride1
                if (y1 < 0 || y1 % g.s1 != 0 || y1 / g.s1 >= g.oH) continue;
                for (int i = -g.kr; i <= g.kr; ++i) {
                    const int x1 = (WHICH == 1 ? x : xo) + g.pad - i - g.md;
                    if (x1 < 0 || x1 % g.s1 != 0 || x1 / g.s1 >= g.oW) continue;
                    acc = __builtin_fmaf(ldf(gb, ((long long)tc * g.oH + y1 / g.s1) * g.oW + x1 / g.s1), ov, acc);
                }
            }
        }
    stf(grad, idx, acc * g.inv_n);
}

CorrCfg make_cfg(int B, int C, int H, int W, int pad, int k, int md, int s1, int s2) {
    CorrCfg g;
    g.B = B; g.C = C; g.H = H; g.W = W; g.pad = pad; g.kr = (k - 1) / 2; g.dr = md / s2; g.ds = 2 * g.dr + 1; g.md = md; g.s1 = s1; g.s2 = s2;
    correlation_output_size(H, W, pad, k, md, s1, &g.oH, &g.oW);
    g.inv_n = 1.0f / (float)(k * k * C);   // the reference divides by nelems (a scalar_t); for fp32 the quotient and this
    return g;                              // product agree to an ulp
}

}  // namespace

// output size of a configuration (correlation_cuda.cc:24-33); false if the configuration is one the reference's kernel
// reads out of bounds for (or has an empty output)
bool correlation_output_size(int H, int W, int pad, int k, int md, int s1, int* oH, int* oW) {
    const int kr = (k - 1) / 2, border = kr + md;
    const int ph = H + 2 * pad - 2 * border, pw = W + 2 * pad - 2 * border;
    *oH = ph > 0 ? (ph + s1 - 1) / s1 : 0;
    *oW = pw > 0 ? (pw + s1 - 1) / s1 : 0;
    return *oH > 0 && *oW > 0;
}

template <typename T>
static hipError_t fwd_t(const T* x1, const T* x2, int B, int C, int H, int W, int pad, int k, int md, int s1, int s2, T* out, hipStream_t st) {
    const CorrCfg g = make_cfg(B, C, H, W, pad, k, md, s1, s2);
    const long long total = (long long)B * g.ds * g.ds * g.oH * g.oW;
    hipLaunchKernelGGL((corr_general_fwd<T>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, x1, x2, out, g);
    return hipGetLastError();
}
template <typename T>
static hipError_t bwd_t(const T* x1, const T* x2, const T* go, int B, int C, int H, int W, int pad, int k, int md, int s1, int s2, T* g1, T* g2,
                        hipStream_t st) {
    const CorrCfg g = make_cfg(B, C, H, W, pad, k, md, s1, s2);
    const long long total = (long long)B * C * H * W;
    const dim3 grid((unsigned)((total + 255) / 256));
    if (g1) hipLaunchKernelGGL((corr_general_bwd<T, 1>), grid, dim3(256), 0, st, x2, go, g1, g);
    if (g2) hipLaunchKernelGGL((corr_general_bwd<T, 2>), grid, dim3(256), 0, st, x1, go, g2, g);
    return hipGetLastError();
}

hipError_t launch_correlation_general_forward(const void* x1, const void* x2, int half, int B, int C, int H, int W, int pad, int k, int md,
                                              int s1, int s2, void* out, hipStream_t st) {
    return half ? fwd_t((const __half*)x1, (const __half*)x2, B, C, H, W, pad, k, md, s1, s2, (__half*)out, st)
                : fwd_t((const float*)x1, (const float*)x2, B, C, H, W, pad, k, md, s1, s2, (float*)out, st);
}
hipError_t launch_correlation_general_backward(const void* x1, const void* x2, const void* go, int half, int B, int C, int H, int W, int pad,
                                               int k, int md, int s1, int s2, void* g1, void* g2, hipStream_t st) {
    return half ? bwd_t((const __half*)x1, (const __half*)x2, (const __half*)go, B, C, H, W, pad, k, md, s1, s2, (__half*)g1, (__half*)g2, st)
                : bwd_t((const float*)x1, (const float*)x2, (const float*)go, B, C, H, W, pad, k, md, s1, s2, (float*)g1, (float*)g2, st);
}

}  // namespace pdepth
