// Direct plane-sweep kernel: per-plane bilinear gather in the reference's op order.
//
// Replaces est_swp_volume_v4 / _back_warp_homo_parallel / img_dis_L{1,2}_pard
// (warping/homography.py:98-135, :170-198, :80-86) and, when logp/depth are requested, the
// log_softmax + dpv_to_depthmap tail (models/packnet.py:394, utils/img_utils.py:52-61).
//
// This is the always-valid path: any pose, any metric, any C/D.  One wave (64 lanes) owns
// 64 consecutive pixels of one batch item; the D per-pixel costs live in LDS as
// cost[k][lane] so the fused softmax/expectation epilogue needs no second pass over HBM and
// the cost/logp stores are 256-byte coalesced rows.  Source taps are gathered from global
// memory (L1/L2); the reference feature vector of the pixel is held in registers.
#include <hip/hip_runtime.h>

#include "geometry.hpp"
#include "kernels.hpp"

namespace pdepth {

// tile_flags == nullptr : block i owns pixels [64 i, 64 i + 64) of the flattened image.
// tile_flags != nullptr : block i owns the 16 wide x 4 tall tile i and runs only if the tiled
//                         kernel flagged that tile (sweep_tiled.hip).
// PACKED: the taps come from the packed copy of the source (channel c of texel p = component c & 3 of float4 plane c >> 2)
// instead of the NCHW tensor; same values, same arithmetic.
template <int METRIC, int CCH, bool MULTI_CHUNK, bool PACKED = false>
__global__ __launch_bounds__(64) void sweep_direct_kernel(SweepArgs a, const int* __restrict__ tile_flags,
                                                          const int* __restrict__ gather_count, int tiles_x, int tiles,
                                                          int flag_value, int item_stride) {
    // the usual case -- no tile was handed over -- costs one scalar load per block
    if (gather_count && *gather_count == 0) return;
    // per-ITEM mode (item_stride != 0: the distance-form kernel's routing, sweep_dist.hip): tile_flags[b * item_stride] != 0 = this
    // batch item is this kernel's, whole; the blocks stride over its groups of 64 pixels
    const bool per_item = item_stride != 0;
    if (per_item && tile_flags[(size_t)blockIdx.y * item_stride] == 0) return;
    extern __shared__ float lds[];
    const int tid = threadIdx.x;
    const int HW = a.H * a.W;
    const int b = blockIdx.y;
    // Flagged mode: a small grid strides over the tiles, so that the usual case -- nothing flagged -- costs a few
    // flag reads per block instead of one (empty) block per tile.
    for (int tile = blockIdx.x; tile < (tile_flags ? tiles : (int)gridDim.x); tile += gridDim.x) {
    int pix;
    bool live;
    if (tile_flags && !per_item) {
        if (tile_flags[b * tiles + tile] != flag_value) continue;  // block-uniform
        const int x = (tile % tiles_x) * 16 + (tid & 15);
        const int y = (tile / tiles_x) * 4 + (tid >> 4);
        live = x < a.W && y < a.H;
        pix = y * a.W + x;
    } else {
        pix = tile * 64 + tid;
        live = pix < HW;
    }
    const int p = live ? pix : HW - 1;  // dead lanes shadow the last pixel, never store

    float* cost = lds;                                  // [D][64]
    float* acc = MULTI_CHUNK ? lds + a.D * 64 : nullptr;   // [D][64], raw per-view sums: level 1 of the cascade, then the total
    float* acc2 = MULTI_CHUNK && a.C >= 256 ? lds + a.D * 128 : nullptr;  // [D][64], level 2 (256 channels and more)
    for (int k = 0; k < a.D; ++k) cost[k * 64 + tid] = 0.0f;

    const float cx = a.cxcy[b * 2 + 0];
    const float cy = a.cxcy[b * 2 + 1];
    const float half_w = (float)a.W / 2.0f;
    const float half_h = (float)a.H / 2.0f;
    const float r0 = a.rays[((size_t)b * 3 + 0) * HW + p];
    const float r1 = a.rays[((size_t)b * 3 + 1) * HW + p];
    const float r2 = a.rays[((size_t)b * 3 + 2) * HW + p];
    const float* refp = a.ref + (size_t)b * a.ref_bstride + p;

    for (int v = 0; v < a.V; ++v) {
        ViewXform xf;
        make_view_xform(a.K + b * 9, a.R + ((size_t)b * a.V + v) * 9,
                        a.t + ((size_t)b * a.V + v) * 3, a.blas_mode, xf);
        float t2a, t2b, t2c;
        ray_term2(xf, r0, r1, r2, t2a, t2b, t2c);
        const float* srcv = PACKED ? reinterpret_cast<const float*>(static_cast<const float4*>(a.packed_src) +
                                                                    ((size_t)b * a.V + v) * ((a.C + 3) / 4 + 2) * HW)
                                   : a.src + (size_t)b * a.src_bstride + (size_t)v * a.src_vstride;

        if (MULTI_CHUNK)
            for (int k = 0; k < a.D; ++k) {
                acc[k * 64 + tid] = 0.0f;
                if (acc2) acc2[k * 64 + tid] = 0.0f;
            }

        for (int c0 = 0; c0 < a.C; c0 += CCH) {
            float rf[CCH];
#pragma unroll
            for (int cc = 0; cc < CCH; ++cc)
                rf[cc] = (c0 + cc < a.C) ? refp[(size_t)(c0 + cc) * HW] : 0.0f;

            for (int k = 0; k < a.D; ++k) {
                float ix, iy;
                plane_sample_pos(xf, t2a, t2b, t2c, a.d_candi[k], cx, cy, half_w, half_h, ix, iy);
                const Footprint f = make_footprint(ix, iy, a.W, a.H);
                const float* s00 = PACKED ? srcv + (size_t)(f.y0 * a.W + f.x0) * 4 : srcv + (size_t)c0 * HW + (f.y0 * a.W + f.x0);
                constexpr int TS = PACKED ? 4 : 1;   // floats between horizontally adjacent texels
                // Sum over the channels in the order of ATen's sum(dim=1) on [D, C, h, w] (SumKernel.cpp: cascade_sum ->
                // vectorized_outer_sum -> multi_row_sum, level_step = 16 up to 2^19 elements): runs of 16 channels are summed
                // from zero (s0), a finished run is added to the level above (s1), 16 runs of that to the next (s2); the
                // left-over channels stay in s0 and the result is (s0 + s1) + s2.  MULTI_CHUNK: s1 (and s2) live in LDS
                // between the chunks (CCH is a multiple of 16: a chunk starts at a run boundary).
                float s0 = 0.0f, s1 = MULTI_CHUNK ? acc[k * 64 + tid] : 0.0f;
#pragma unroll
                for (int cc = 0; cc < CCH; ++cc) {
                    if (c0 + cc < a.C) {  // wave-uniform
                        const int c = c0 + cc;
                        const float* s = PACKED ? s00 + (size_t)(c >> 2) * HW * 4 + (c & 3) : s00 + (size_t)cc * HW;
                        const float vnw = (f.mask & 1u) ? s[0] : 0.0f;
                        const float vne = (f.mask & 2u) ? s[TS] : 0.0f;
                        const float vsw = (f.mask & 4u) ? s[a.W * TS] : 0.0f;
                        const float vse = (f.mask & 8u) ? s[(a.W + 1) * TS] : 0.0f;
                        float val = vnw * f.nw;
                        val = __builtin_fmaf(vne, f.ne, val);
                        val = __builtin_fmaf(vsw, f.sw, val);
                        val = __builtin_fmaf(vse, f.se, val);
                        const float diff = val - rf[cc];
                        s0 = s0 + (METRIC == 0 ? diff * diff : fabsf(diff));
                        if ((cc & 15) == 15) {
                            s1 = s1 + s0;
                            s0 = 0.0f;
                            if (MULTI_CHUNK && (c & 255) == 255) {   // (acc2 != nullptr: C >= 256)
                                acc2[k * 64 + tid] = acc2[k * 64 + tid] + s1;
                                s1 = 0.0f;
                            }
                        }
                    }
                }
                if (MULTI_CHUNK) {
                    if (c0 + CCH < a.C) acc[k * 64 + tid] = s1;                      // (s0 == 0: whole runs only)
                    else acc[k * 64 + tid] = acc2 ? (s0 + s1) + acc2[k * 64 + tid] : s0 + s1;
                } else {
                    cost[k * 64 + tid] = cost[k * 64 + tid] + (s0 + s1) / a.sigma;
                }
            }
        }
        if (MULTI_CHUNK)
            for (int k = 0; k < a.D; ++k)
                cost[k * 64 + tid] = cost[k * 64 + tid] + acc[k * 64 + tid] / a.sigma;
    }

    // ---- epilogue: cost store, log-softmax over D, expectation --------------------------
    if (a.cost_out && live) {
        float* o = a.cost_out + (size_t)b * a.D * HW + pix;
        for (int k = 0; k < a.D; ++k) o[(size_t)k * HW] = cost[k * 64 + tid];
    }
    if (a.logp_out || a.depth_out) {
        float m = -INFINITY;
        for (int k = 0; k < a.D; ++k) m = fmaxf(m, cost[k * 64 + tid]);
        float s = 0.0f;
        for (int k = 0; k < a.D; ++k) s = s + expf(cost[k * 64 + tid] - m);
        const float ls = logf(s);
        // expectation: torch.sum(d * exp(logp), dim=0) (utils/img_utils.py:59), in the order of ATen's cascade (see above)
        float e0 = 0.0f, e1 = 0.0f, e2 = 0.0f;
        float* o = a.logp_out ? a.logp_out + (size_t)b * a.D * HW + pix : nullptr;
        for (int k = 0; k < a.D; ++k) {
            const float lp = (cost[k * 64 + tid] - m) - ls;
            if (o && live) o[(size_t)k * HW] = lp;
            e0 = e0 + a.d_candi[k] * expf(lp);
            if ((k & 15) == 15) {
                e1 = e1 + e0;
                e0 = 0.0f;
                if ((k & 255) == 255) {
                    e2 = e2 + e1;
                    e1 = 0.0f;
                }
            }
        }
        if (a.depth_out && live) a.depth_out[(size_t)b * HW + pix] = (e0 + e1) + e2;
    }
    __syncthreads();   // (one wave per block: orders this tile's LDS reads before the next tile's writes)
    }  // tiles
}

// per-item mode: blocks per batch item (each leaves after one flag read where its item is not routed: the usual case, paid by
// every NCHW call of the default kernel)
#ifndef PDEPTH_ROUTE_GRID
#define PDEPTH_ROUTE_GRID 2048
#endif

template <int METRIC>
static hipError_t launch_metric(const SweepArgs& a, const int* tile_flags, const int* gather_count, int tiles_x, int tiles,
                                hipStream_t stream, int flag_value = 1, int item_stride = 0) {
    const int HW = a.H * a.W;
    if (item_stride) tiles = (HW + 63) / 64;
    dim3 grid(item_stride ? (tiles < PDEPTH_ROUTE_GRID ? tiles : PDEPTH_ROUTE_GRID) : tile_flags ? (tiles < 256 ? tiles : 256) : (HW + 63) / 64, a.B);
    // no NCHW source (packed-source entry): the taps come from the packed copy
    const bool packed = a.src == nullptr;
    if (packed && a.packed_src == nullptr) return hipErrorInvalidValue;
    auto go = [&](auto kern, size_t lds) -> hipError_t {
        if (lds > 64 * 1024) {
            hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
        }
        hipLaunchKernelGGL(kern, grid, dim3(64), lds, stream, a, tile_flags, gather_count, tiles_x, tiles, flag_value, item_stride);
        return hipGetLastError();
    };
    if (a.C <= 68) {
        const size_t lds = (size_t)a.D * 64 * sizeof(float);
        return packed ? go(sweep_direct_kernel<METRIC, 68, false, true>, lds) : go(sweep_direct_kernel<METRIC, 68, false, false>, lds);
    }
    const size_t lds = (size_t)a.D * 64 * sizeof(float) * (a.C >= 256 ? 3 : 2);
    return packed ? go(sweep_direct_kernel<METRIC, 32, true, true>, lds) : go(sweep_direct_kernel<METRIC, 32, true, false>, lds);
}

hipError_t launch_sweep_direct(const SweepArgs& a, hipStream_t stream) {
    return a.metric == 0 ? launch_metric<0>(a, nullptr, nullptr, 0, 0, stream)
                         : launch_metric<1>(a, nullptr, nullptr, 0, 0, stream);
}

hipError_t launch_sweep_direct_flagged(const SweepArgs& a, const int* tile_flags, const int* gather_count, int tiles_x,
                                       int tiles, hipStream_t stream, int flag_value) {
    return a.metric == 0 ? launch_metric<0>(a, tile_flags, gather_count, tiles_x, tiles, stream, flag_value)
                         : launch_metric<1>(a, tile_flags, gather_count, tiles_x, tiles, stream, flag_value);
}

// the batch items with item_flags[b * item_stride] != 0, whole (the other items' blocks leave at once)
hipError_t launch_sweep_direct_items(const SweepArgs& a, const int* item_flags, int item_stride, hipStream_t stream) {
    return a.metric == 0 ? launch_metric<0>(a, item_flags, nullptr, 0, 0, stream, 1, item_stride)
                         : launch_metric<1>(a, item_flags, nullptr, 0, 0, stream, 1, item_stride);
}

// Largest D the direct kernel can hold in LDS (two arrays in the chunked variant, three from 256 channels on).
int sweep_direct_max_planes(int C) { return C <= 68 ? 512 : (C < 256 ? 256 : 170); }  // 128 KB of LDS

}  // namespace pdepth
