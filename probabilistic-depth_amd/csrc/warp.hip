// Diagonal feature warp + coordinate dump.
//
// warp_feature (warping/homography.py:137-168): the reference warps every channel of a
// view with every depth plane ([D,C,h,w]) and keeps only the diagonal [i,i] -- channel i
// sampled with plane i.  This kernel samples exactly that diagonal: one bilinear gather per
// (view, plane, pixel), 1/D of the reference's work, 8*h*w*V*D algorithmic bytes.
#include <hip/hip_runtime.h>

#include "geometry.hpp"
#include "kernels.hpp"

namespace pdepth {

// grid (pixel blocks, V * nchunk, B): a thread takes planes chunk, chunk + nchunk, ... of its pixel -- on the model's small
// maps (64x128) one thread per pixel walking all D planes leaves the chip a block per CU and a chain of dependent gathers.
__global__ __launch_bounds__(256) void warp_feature_kernel(SweepArgs a, int nchunk, float* __restrict__ out) {
    const int HW = a.H * a.W;
    const int pix = blockIdx.x * 256 + threadIdx.x;
    if (pix >= HW) return;
    const int v = blockIdx.y / nchunk, chunk = blockIdx.y - v * nchunk;
    const int b = blockIdx.z;
    ViewXform xf;
    make_view_xform(a.K + b * 9, a.R + ((size_t)b * a.V + v) * 9, a.t + ((size_t)b * a.V + v) * 3, a.blas_mode, xf);
    float t2a, t2b, t2c;
    ray_term2(xf, a.rays[((size_t)b * 3 + 0) * HW + pix], a.rays[((size_t)b * 3 + 1) * HW + pix],
              a.rays[((size_t)b * 3 + 2) * HW + pix], t2a, t2b, t2c);
    const float cx = a.cxcy[b * 2 + 0], cy = a.cxcy[b * 2 + 1];
    const float half_w = (float)a.W / 2.0f, half_h = (float)a.H / 2.0f;
    const float rcx = refined_rcp(cx), rcy = refined_rcp(cy);
    const float* srcv = a.src + (size_t)b * a.src_bstride + (size_t)v * a.src_vstride;
    float* o = out + (((size_t)b * a.V + v) * a.D) * HW + pix;
    for (int k = chunk; k < a.D; k += nchunk) {
        float ix, iy;
        // (the fast divide chain is bit-identical to the IEEE one wherever a tap can land inside the image: geometry.hpp)
        if (a.fast_div) plane_sample_pos_fast(xf, t2a, t2b, t2c, a.d_candi[k], cx, cy, rcx, rcy, half_w, half_h, ix, iy);
        else plane_sample_pos(xf, t2a, t2b, t2c, a.d_candi[k], cx, cy, half_w, half_h, ix, iy);
        const Footprint f = make_footprint(ix, iy, a.W, a.H);
        const float* s = srcv + (size_t)k * HW + (f.y0 * a.W + f.x0);
        const float vnw = (f.mask & 1u) ? s[0] : 0.0f;
        const float vne = (f.mask & 2u) ? s[1] : 0.0f;
        const float vsw = (f.mask & 4u) ? s[a.W] : 0.0f;
        const float vse = (f.mask & 8u) ? s[a.W + 1] : 0.0f;
        float val = vnw * f.nw;
        val = __builtin_fmaf(vne, f.ne, val);
        val = __builtin_fmaf(vsw, f.sw, val);
        val = __builtin_fmaf(vse, f.se, val);
        o[(size_t)k * HW] = val;
    }
}

__global__ __launch_bounds__(256) void sample_coords_kernel(SweepArgs a, float* __restrict__ oix,
                                                            float* __restrict__ oiy) {
    const int HW = a.H * a.W;
    const int pix = blockIdx.x * 256 + threadIdx.x;
    if (pix >= HW) return;
    const int v = blockIdx.y;
    const int b = blockIdx.z;
    ViewXform xf;
    make_view_xform(a.K + b * 9, a.R + ((size_t)b * a.V + v) * 9, a.t + ((size_t)b * a.V + v) * 3, a.blas_mode, xf);
    float t2a, t2b, t2c;
    ray_term2(xf, a.rays[((size_t)b * 3 + 0) * HW + pix], a.rays[((size_t)b * 3 + 1) * HW + pix],
              a.rays[((size_t)b * 3 + 2) * HW + pix], t2a, t2b, t2c);
    const float cx = a.cxcy[b * 2 + 0], cy = a.cxcy[b * 2 + 1];
    const float half_w = (float)a.W / 2.0f, half_h = (float)a.H / 2.0f;
    const size_t base = (((size_t)b * a.V + v) * a.D) * HW + pix;
    const float rcx = refined_rcp(cx), rcy = refined_rcp(cy);
    for (int k = 0; k < a.D; ++k) {
        float ix, iy;
        if (a.fast_div) plane_sample_pos_fast(xf, t2a, t2b, t2c, a.d_candi[k], cx, cy, rcx, rcy, half_w, half_h, ix, iy);
        else plane_sample_pos(xf, t2a, t2b, t2c, a.d_candi[k], cx, cy, half_w, half_h, ix, iy);
        oix[base + (size_t)k * HW] = ix;
        oiy[base + (size_t)k * HW] = iy;
    }
}

hipError_t launch_warp_feature(const SweepArgs& a, float* out, hipStream_t stream) {
    const int pixblocks = (a.H * a.W + 255) / 256;
    long long nchunk = 2048 / ((long long)pixblocks * a.V * a.B);   // at least ~8 blocks per CU
    nchunk = nchunk < 1 ? 1 : nchunk > 16 ? 16 : nchunk;
    if (nchunk > a.D) nchunk = a.D;
    if ((long long)a.V * nchunk > 65535) nchunk = 1;
    dim3 grid(pixblocks, a.V * (int)nchunk, a.B);
    hipLaunchKernelGGL(warp_feature_kernel, grid, dim3(256), 0, stream, a, (int)nchunk, out);
    return hipGetLastError();
}

hipError_t launch_sample_coords(const SweepArgs& a, float* ix, float* iy, hipStream_t stream) {
    dim3 grid((a.H * a.W + 255) / 256, a.V, a.B);
    hipLaunchKernelGGL(sample_coords_kernel, grid, dim3(256), 0, stream, a, ix, iy);
    return hipGetLastError();
}

}  // namespace pdepth
