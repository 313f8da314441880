// Uncertainty-field collapse of a depth probability volume: pdepth_ufield_f32.
//
// Replaces gen_ufield (utils/img_utils.py:268-358, cfgx branch; consumer compute_unc_field :178-181 from
// trainer/default_trainer.py:243-244): the [D,H,W] volume of one view is collapsed to a [D,W] "bird's eye" plane -- per
// image column the mean of the depth distributions of the pixels that lie in a height band above the ground -- plus
// the depth map masked to those pixels.  The reference makes ~12 passes over [D,H,W] tensors (two grid_samples of the
// whole volume, exp, repeat of the mask over D, multiply, sum); here the volume is read twice (expectation, collapse)
// and everything in between works on [H,W] maps:
//
//   1. depth_pred = E[d] of the volume                                   (dpv.hip, launch_dpv_expect)
//   2. ufield_mask_kernel, one block per column: the depth map of the volume shifted by `unc_ang` rows IS the shifted
//      depth map (dpv_to_depthmap works per pixel), so only depth_pred is gathered through the reference's nearest
//      sampling grid; height-band / range mask of the shifted points, validity mask, "quash" to the nearest surface
//      of the column (block-wide min), column counts ax;
//   3. ufield_collapse_kernel: plane[d,x] = sum_y p(d,y,x) * mask_back(y,x) / ax(x), mask_back = the mask sampled
//      through the inverse shift; lanes along x (coalesced rows), 4 planes per block share the mask reads.
//
// The sampling grid is reproduced literally: the reference builds it with the (size - 1) convention
// (convert_flowfield, :170-176) but samples with grid_sample's default align_corners=False, so it is NOT a pure row
// shift -- e.g. for even W the last column rounds out of the image and reads padding zeros (which, for a log-DPV, the
// reference then exponentiates to ones).  nearest = round-half-to-even of fma(g + 1, size / 2, -0.5), as ATen's CPU
// kernel evaluates it (geometry.hpp has the same un-normalisation for the bilinear sampler).
#include <hip/hip_runtime.h>

#include <cstdint>

#include "kernels.hpp"

namespace pdepth {

namespace {

// source index of the reference's nearest sampling for destination index i, pixel offset `shift`; -1 = outside
__device__ __forceinline__ int nearest_src(int i, float shift, int size, bool sampled) {
    if (!sampled) return i;   // unc_ang == 0: the reference clones instead of sampling (:301)
    const float step = 2.0f / (float)(size - 1);
    const float g = (-1.0f + (float)i * step) - shift * step;          // convert_flowfield, op for op
    const float pos = __builtin_fmaf(g + 1.0f, (float)size / 2.0f, -0.5f);
    const float r = rintf(pos);                                        // round half to even, like Vec::round
    return (r >= 0.0f && r < (float)size) ? (int)r : -1;
}

}  // namespace

// grid (W, B), block 256.  zero_mask [B,H,W], ax [B,W].
__global__ __launch_bounds__(256) void ufield_mask_kernel(const float* __restrict__ depth_pred, const float* __restrict__ intr,
                                                          const float* __restrict__ mask, int H, int W, float pshift,
                                                          float zstart, float zend, float mind, int quash, float oob_depth,
                                                          float* __restrict__ zero_mask, float* __restrict__ ax) {
    __shared__ float s_red[256];
    const int x = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const float cy = intr[b * 9 + 5], fy = intr[b * 9 + 4];
    const float maxd = 100.0f;
    const bool sampled = pshift != 0.0f;
    const int sx = nearest_src(x, 0.0f, W, sampled);
    const float* dp = depth_pred + (size_t)b * H * W;
    float colmin = INFINITY;
    // pass 1: band mask and the column minimum of the masked depth
    for (int y = tid; y < H; y += 256) {
        const int sy = nearest_src(y, pshift, H, sampled);
        const bool inb = sx >= 0 && sy >= 0;
        const float d = inb ? dp[(size_t)sy * W + sx] : oob_depth;
        const float yf = ((float)y - cy) / fy;
        const float Y = yf * d;
        float zm = !((Y > zend) || (Y < zstart) || (d > maxd - 1.0f) || (d < mind)) ? 1.0f : 0.0f;
        if (mask) zm = zm * (inb ? mask[(size_t)b * H * W + (size_t)sy * W + sx] : 0.0f);
        float cleaned = d * zm;
        if (cleaned == 0.0f) cleaned = 1000.0f;
        colmin = fminf(colmin, cleaned);
        zero_mask[(size_t)b * H * W + (size_t)y * W + x] = zm;   // (provisional: multiplied by the quash mask below)
    }
    s_red[tid] = colmin;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) s_red[tid] = fminf(s_red[tid], s_red[tid + s]);
        __syncthreads();
    }
    colmin = s_red[0];
    __syncthreads();
    // pass 2: quash everything that is not within 1 m of the column's nearest surface; count
    float cnt = 0.0f;
    for (int y = tid; y < H; y += 256) {
        const int sy = nearest_src(y, pshift, H, sampled);
        const bool inb = sx >= 0 && sy >= 0;
        const float d = inb ? dp[(size_t)sy * W + sx] : oob_depth;
        float zm = zero_mask[(size_t)b * H * W + (size_t)y * W + x];
        float cleaned = d * zm;
        if (cleaned == 0.0f) cleaned = 1000.0f;
        const bool keep = !quash || ((cleaned > colmin - 1.0f) && (cleaned < colmin + 1.0f));
        zm = zm * (keep ? 1.0f : 0.0f);
        zero_mask[(size_t)b * H * W + (size_t)y * W + x] = zm;
        cnt += zm;
    }
    s_red[tid] = cnt;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) s_red[tid] += s_red[tid + s];
        __syncthreads();
    }
    if (tid == 0) ax[b * W + x] = s_red[0];
}

// grid (ceil(W/64), ceil(D/4), B), block (64, 8): lane = column, threadIdx.y = eighth of the rows; every thread
// accumulates 4 planes over its rows (4 independent loads per row, 4 rows unrolled, the mask read once for the 4
// planes: the kernel lives on loads in flight), the eighths are combined through LDS in row order.
template <bool BV_LOG>
__global__ __launch_bounds__(512) void ufield_collapse_kernel(const float* __restrict__ dpv, const float* __restrict__ depth_pred,
                                                              const float* __restrict__ zero_mask, const float* __restrict__ ax,
                                                              int D, int H, int W, float pshift, float* __restrict__ plane,
                                                              float* __restrict__ depth_zero) {
    __shared__ float s_part[8][4][64];
    const int x = blockIdx.x * 64 + threadIdx.x, k0 = blockIdx.y * 4, b = blockIdx.z, seg = threadIdx.y;
    const bool col = x < W;
    const bool sampled = pshift != 0.0f;
    const int sx = col ? nearest_src(x, 0.0f, W, sampled) : -1;
    const float* zm = zero_mask + (size_t)b * H * W;
    const size_t HW = (size_t)H * W;
    const float* v = dpv + ((size_t)b * D + k0) * HW + x;
    const int rows = (H + 7) / 8, y_lo = seg * rows, y_hi = min(H, y_lo + rows);
    float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    if (col) {
#pragma unroll 4
        for (int y = y_lo; y < y_hi; ++y) {
            const int sy = nearest_src(y, -pshift, H, sampled);   // the mask is shifted back (flowfield_inv)
            const float m = (sx >= 0 && sy >= 0) ? zm[(size_t)sy * W + sx] : 0.0f;
            float p[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) p[j] = (k0 + j < D) ? v[j * HW + (size_t)y * W] : 0.0f;
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = acc[j] + (BV_LOG ? expf(p[j]) : p[j]) * m;
            if (k0 == 0) depth_zero[(size_t)b * HW + (size_t)y * W + x] = depth_pred[(size_t)b * HW + (size_t)y * W + x] * m;
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) s_part[seg][j][threadIdx.x] = acc[j];
    __syncthreads();
    const int j = threadIdx.y;   // thread (lane, j) finishes plane k0 + j of column x
    if (col && j < 4 && k0 + j < D) {
        float sum = s_part[0][j][threadIdx.x];
#pragma unroll
        for (int q = 1; q < 8; ++q) sum = sum + s_part[q][j][threadIdx.x];
        plane[((size_t)b * D + k0 + j) * W + x] = sum / ax[b * W + x];   // 0 / 0 = NaN where no pixel of the column qualifies, as in the reference
    }
}

// The same collapse for W % 4 == 0 (every row 16-byte aligned): lane = 4 neighbouring columns (16-byte loads: a wave reads
// 1 KB of a row per instruction instead of 256 B), threadIdx.y = sixteenth of the rows, 2 planes per block.  Same sums in
// the same order per column (rows ascending inside a segment, segments ascending), so the results match the scalar kernel's
// whenever H splits into the same segments; in general they differ by the rounding of a different association.
// grid (ceil(W/256), ceil(D/2), B), block (64, 16).
template <bool BV_LOG>
__global__ __launch_bounds__(1024) void ufield_collapse_vec4_kernel(const float* __restrict__ dpv, const float* __restrict__ depth_pred,
                                                                    const float* __restrict__ zero_mask, const float* __restrict__ ax,
                                                                    int D, int H, int W, float pshift, float* __restrict__ plane,
                                                                    float* __restrict__ depth_zero) {
    __shared__ float4 s_part[16][2][64];
    const int x = (blockIdx.x * 64 + threadIdx.x) * 4, k0 = blockIdx.y * 2, b = blockIdx.z, seg = threadIdx.y;
    const bool col = x < W;
    const bool sampled = pshift != 0.0f;
    int sx[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) sx[i] = col ? nearest_src(x + i, 0.0f, W, sampled) : -1;
    const bool straight = sx[0] == x && sx[1] == x + 1 && sx[2] == x + 2 && sx[3] == x + 3;   // (the mask row segment is one 16-byte load)
    const float* zm = zero_mask + (size_t)b * H * W;
    const size_t HW = (size_t)H * W;
    const float* v = dpv + ((size_t)b * D + k0) * HW + x;
    const int rows = (H + 15) / 16, y_lo = seg * rows, y_hi = min(H, y_lo + rows);
    float4 acc[2] = {make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f)};
    const bool two = k0 + 1 < D;
    if (col) {
#pragma unroll 4
        for (int y = y_lo; y < y_hi; ++y) {
            const int sy = nearest_src(y, -pshift, H, sampled);   // the mask is shifted back (flowfield_inv)
            float4 m = make_float4(0.f, 0.f, 0.f, 0.f);
            if (sy >= 0) {
                if (straight) m = *reinterpret_cast<const float4*>(zm + (size_t)sy * W + x);
                else {
                    m.x = sx[0] >= 0 ? zm[(size_t)sy * W + sx[0]] : 0.0f; m.y = sx[1] >= 0 ? zm[(size_t)sy * W + sx[1]] : 0.0f;
                    m.z = sx[2] >= 0 ? zm[(size_t)sy * W + sx[2]] : 0.0f; m.w = sx[3] >= 0 ? zm[(size_t)sy * W + sx[3]] : 0.0f;
                }
            }
            const float4 p0 = *reinterpret_cast<const float4*>(v + (size_t)y * W);
            const float4 p1 = two ? *reinterpret_cast<const float4*>(v + HW + (size_t)y * W) : make_float4(0.f, 0.f, 0.f, 0.f);
            auto val = [](float p) { return BV_LOG ? expf(p) : p; };
            acc[0].x = acc[0].x + val(p0.x) * m.x; acc[0].y = acc[0].y + val(p0.y) * m.y;
            acc[0].z = acc[0].z + val(p0.z) * m.z; acc[0].w = acc[0].w + val(p0.w) * m.w;
            acc[1].x = acc[1].x + val(p1.x) * m.x; acc[1].y = acc[1].y + val(p1.y) * m.y;
            acc[1].z = acc[1].z + val(p1.z) * m.z; acc[1].w = acc[1].w + val(p1.w) * m.w;
            if (k0 == 0) {
                const float4 dpd = *reinterpret_cast<const float4*>(depth_pred + (size_t)b * HW + (size_t)y * W + x);
                *reinterpret_cast<float4*>(depth_zero + (size_t)b * HW + (size_t)y * W + x) = make_float4(dpd.x * m.x, dpd.y * m.y, dpd.z * m.z, dpd.w * m.w);
            }
        }
    }
    s_part[seg][0][threadIdx.x] = acc[0];
    s_part[seg][1][threadIdx.x] = acc[1];
    __syncthreads();
    const int j = threadIdx.y;   // thread (lane, j) finishes plane k0 + j of its 4 columns
    if (col && j < 2 && k0 + j < D) {
        float4 sum = s_part[0][j][threadIdx.x];
#pragma unroll
        for (int q = 1; q < 16; ++q) {
            const float4 t = s_part[q][j][threadIdx.x];
            sum.x = sum.x + t.x; sum.y = sum.y + t.y; sum.z = sum.z + t.z; sum.w = sum.w + t.w;
        }
        const float4 a4 = *reinterpret_cast<const float4*>(ax + b * W + x);
        // 0 / 0 = NaN where no pixel of the column qualifies, as in the reference
        *reinterpret_cast<float4*>(plane + ((size_t)b * D + k0 + j) * W + x) = make_float4(sum.x / a4.x, sum.y / a4.y, sum.z / a4.z, sum.w / a4.w);
    }
}

size_t ufield_workspace_bytes(int B, int H, int W) { return ((size_t)B * (2 * (size_t)H * W + W) * sizeof(float) + 255) & ~(size_t)255; }

hipError_t launch_ufield(const float* dpv, const float* d_candi, const float* intr, const float* mask, int B, int D, int H,
                         int W, int bv_log, float unc_ang, float zstart, float zend, float mind, int quash, float oob_depth,
                         float* plane, float* depth_zero, void* workspace, hipStream_t stream) {
    float* depth_pred = static_cast<float*>(workspace);
    float* zero_mask = depth_pred + (size_t)B * H * W;
    float* ax = zero_mask + (size_t)B * H * W;
    hipError_t e = launch_dpv_expect(dpv, d_candi, B, D, H, W, bv_log, depth_pred, stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(ufield_mask_kernel, dim3(W, B), dim3(256), 0, stream, depth_pred, intr, mask, H, W, unc_ang, zstart, zend,
                       mind, quash, oob_depth, zero_mask, ax);
    const bool aligned = W % 4 == 0 && ((reinterpret_cast<uintptr_t>(dpv) | reinterpret_cast<uintptr_t>(plane) | reinterpret_cast<uintptr_t>(depth_zero) |
                                         reinterpret_cast<uintptr_t>(workspace)) & 15) == 0;
    if (aligned) {
        dim3 grid4((W + 255) / 256, (D + 1) / 2, B), block4(64, 16);
        if (bv_log)
            hipLaunchKernelGGL(ufield_collapse_vec4_kernel<true>, grid4, block4, 0, stream, dpv, depth_pred, zero_mask, ax, D, H, W, unc_ang, plane, depth_zero);
        else
            hipLaunchKernelGGL(ufield_collapse_vec4_kernel<false>, grid4, block4, 0, stream, dpv, depth_pred, zero_mask, ax, D, H, W, unc_ang, plane, depth_zero);
        return hipGetLastError();
    }
    dim3 grid((W + 63) / 64, (D + 3) / 4, B), block(64, 8);
    if (bv_log)
        hipLaunchKernelGGL(ufield_collapse_kernel<true>, grid, block, 0, stream, dpv, depth_pred, zero_mask, ax, D, H, W, unc_ang, plane, depth_zero);
    else
        hipLaunchKernelGGL(ufield_collapse_kernel<false>, grid, block, 0, stream, dpv, depth_pred, zero_mask, ax, D, H, W, unc_ang, plane, depth_zero);
    return hipGetLastError();
}

}  // namespace pdepth
