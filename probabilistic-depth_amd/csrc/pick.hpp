// "Do the epipolar lines of view 0 run along the source rows?" -- one decision per batch item from four probe pixels: does the
// sample move by more than half a source row between the first and the last depth plane?  (A rectified stereo pair: no.  A
// forward motion: yes.)  Used twice: by the matrix-pipe sweep kernel for the shape of its pixel sub-blocks (sweep_mfma.hip),
// and by the pre-pass of ALGO_AUTO to choose, on the device, between that kernel and the tiled band kernel for the one shape
// class where the faster of the two depends on the pose (single view, at most 64 planes, large image): the host cannot look
// at the poses without a synchronisation, so both kernels are launched and the one not chosen leaves at once.
#pragma once
#include <hip/hip_runtime.h>

#include "geometry.hpp"
#include "kernels.hpp"

namespace pdepth {

// one probe pixel (pr = 0..3) of batch item b: does the sample move by more than half a source row over the depth range?
__device__ __forceinline__ bool epipolar_probe_is_steep(const SweepArgs& a, int b, int pr) {
    ViewXform xf;
    make_view_xform(a.K + b * 9, a.R + (size_t)b * a.V * 9, a.t + (size_t)b * a.V * 3, a.blas_mode, xf);
    const float cx = a.cxcy[b * 2 + 0], cy = a.cxcy[b * 2 + 1];
    const float rcx = refined_rcp(cx), rcy = refined_rcp(cy);
    const float half_w = (float)a.W / 2.0f, half_h = (float)a.H / 2.0f;
    const size_t HW = (size_t)a.H * a.W;
    const int px = (pr & 1) ? (7 * a.W) / 8 : a.W / 8, py = (pr & 2) ? (7 * a.H) / 8 : a.H / 8, pc = py * a.W + px;
    float t2a, t2b, t2c, ix0, iy0, ix1, iy1;
    ray_term2(xf, a.rays[((size_t)b * 3 + 0) * HW + pc], a.rays[((size_t)b * 3 + 1) * HW + pc], a.rays[((size_t)b * 3 + 2) * HW + pc],
              t2a, t2b, t2c);
    plane_sample_pos_fast(xf, t2a, t2b, t2c, a.d_candi[0], cx, cy, rcx, rcy, half_w, half_h, ix0, iy0);
    plane_sample_pos_fast(xf, t2a, t2b, t2c, a.d_candi[a.D - 1], cx, cy, rcx, rcy, half_w, half_h, ix1, iy1);
    return fabsf(iy1 - iy0) > 0.5f;   // (NaN: not steep)
}

__device__ __forceinline__ bool epipolar_lines_are_flat(const SweepArgs& a, int b) {
    ViewXform xf;
    make_view_xform(a.K + b * 9, a.R + (size_t)b * a.V * 9, a.t + (size_t)b * a.V * 3, a.blas_mode, xf);
    const float cx = a.cxcy[b * 2 + 0], cy = a.cxcy[b * 2 + 1];
    const float rcx = refined_rcp(cx), rcy = refined_rcp(cy);
    const float half_w = (float)a.W / 2.0f, half_h = (float)a.H / 2.0f;
    const float d0 = a.d_candi[0], d1 = a.d_candi[a.D - 1];
    const size_t HW = (size_t)a.H * a.W;
    bool steep = false;
    for (int pr = 0; pr < 4; ++pr) {
        const int px = (pr & 1) ? (7 * a.W) / 8 : a.W / 8, py = (pr & 2) ? (7 * a.H) / 8 : a.H / 8, pc = py * a.W + px;
        float t2a, t2b, t2c, ix0, iy0, ix1, iy1;
        ray_term2(xf, a.rays[((size_t)b * 3 + 0) * HW + pc], a.rays[((size_t)b * 3 + 1) * HW + pc], a.rays[((size_t)b * 3 + 2) * HW + pc],
                  t2a, t2b, t2c);
        plane_sample_pos_fast(xf, t2a, t2b, t2c, d0, cx, cy, rcx, rcy, half_w, half_h, ix0, iy0);
        plane_sample_pos_fast(xf, t2a, t2b, t2c, d1, cx, cy, rcx, rcy, half_w, half_h, ix1, iy1);
        steep = steep || fabsf(iy1 - iy0) > 0.5f;   // (NaN: not steep)
    }
    return !steep;
}

// thread `tid` of ONE block: queue[PICK_SLOT] = 1 if every batch item's lines are flat (else it stays 0, as cleared)
__device__ __forceinline__ void pick_for_launch(const SweepArgs& a, int* queue, int tid, int nthreads) {
    bool all_flat = true;
    for (int b = tid; b < a.B; b += nthreads) all_flat = all_flat && epipolar_lines_are_flat(a, b);
    const unsigned long long bad = __builtin_amdgcn_ballot_w64(!all_flat);
    __shared__ int s_bad;
    if (tid == 0) s_bad = 0;
    __syncthreads();
    if (bad != 0ull && (tid & 63) == 0) atomicOr(&s_bad, 1);
    __syncthreads();
    if (tid == 0 && s_bad == 0) queue[PICK_SLOT] = PICK_MFMA;
}

}  // namespace pdepth
