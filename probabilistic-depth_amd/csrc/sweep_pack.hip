// Pre-pass of the packed-source sweep kernels: channel statistics of the source view, then the source views in the
// staging layout the sweep kernels load 16 bytes at a time, plus the workspace bookkeeping they share.
//
// Workspace of a sweep call (pdepth_sweep_workspace_bytes):
//     [tile flags: one int per (batch item, 16x4 tile)]      tiles handed to the gather kernel (LDS-tiled kernel only)
//     [64 ints]                                              queue counters of the persistent kernels + the slots of kernels.hpp
//     [packed source: B*V x (ceil(C/4) + 2) x H x W float4]  planes g < ceil(C/4): channels 4g..4g+3 of every texel, minus
//                                                            mu[c]; then the two Gram planes
//     [tile list]                                            (lab builds: the cell-list kernels)
//     [statistics: B x STATS_STRIDE floats]                  mu[c] (the constant subtracted per channel; zeros = not centred)
//                                                            at +0, var[c] at +STATS_VAR
//
// Mean-centring (sweep_corr.hip says why): mu[b][c] = mean of channel c over a sample of 8 rows of source view 0 of
// item b -- an estimate is all it takes, the correlation form is exact for ANY constant; what matters is that the
// residual offset is small against the spread.  Consumers that do not centre (the LDS-tiled kernel: direct form on the
// near planes) get mu = 0 and the layout is bit for bit the uncentred one; for those the pre-pass also raises
// NONCENTRED_SLOT when sum mu^2 > sum var / 2, and the tiled kernel then evaluates every plane directly.
#include <hip/hip_runtime.h>

#include "dist_layout.hpp"
#include "kernels.hpp"
#include "pack_body.hpp"
#include "pick.hpp"
#include "stats_body.hpp"

namespace pdepth {

namespace {

constexpr int TW = 16, TH = 4;
using namespace stats_body;

// flags != nullptr: the call's workspace bookkeeping is done here (the sweep kernel that packs the source itself has no
// pack kernel in front of it): tile flags and queue slots cleared, the pack counters of the sweep kernel zeroed.
__global__ __launch_bounds__(256) void feature_stats_kernel(const float* __restrict__ src, long long bstride, long long vstride, int V,
                                                            const float* __restrict__ ref, long long ref_bstride, int C, int H, int W,
                                                            float* __restrict__ stats, int centre, int* __restrict__ flags, int nflags,
                                                            int* __restrict__ pack_ctr, int nctr) {
    const int c = blockIdx.x, b = blockIdx.y;
    if (flags) {
        // (nflags covers the tile flags and the 64 queue ints behind them)
        for (int i = (b * gridDim.x + c) * 256 + threadIdx.x; i < nflags; i += gridDim.x * gridDim.y * 256) flags[i] = 0;
        if (c == 0 && b == 0)
            for (int i = threadIdx.x; i < nctr; i += 256) pack_ctr[i] = 0;
    }
    float* st = stats + (size_t)b * STATS_STRIDE;
    if (c == 0 && threadIdx.x < STATS_VAR - C && C + (int)threadIdx.x < STATS_VAR) {   // channels beyond C
        st[C + threadIdx.x] = 0.0f; st[STATS_VAR + C + threadIdx.x] = 0.0f; st[STATS_OFF + C + threadIdx.x] = 0.0f;
        st[STATS_AMAX + C + threadIdx.x] = 0.0f; st[STATS_LAG + C + threadIdx.x] = 0.0f;
    }
    if (c == 0 && threadIdx.x < STATS_NFLAG) reinterpret_cast<int*>(st + STATS_FLAGS)[threadIdx.x] = 0;
    if (c >= STATS_VAR) return;   // (only the first 80 channels are recorded: the centring kernels take C <= 72)
    channel_stats<false>(src + (size_t)b * bstride + (size_t)c * H * W, vstride, V, ref ? ref + (size_t)b * ref_bstride + (size_t)c * H * W : nullptr,
                         H, W, 1, W, st + c, st + STATS_VAR + c, centre);
}

// the same for the encoder epilogue: feat [B*(V+1), Cf, H, W], rgb [B*(V+1), 3, H*rate, W*rate]; source view 0 of item b
__global__ __launch_bounds__(256) void view_stats_kernel(const float* __restrict__ feat, const float* __restrict__ rgb, int V, int Cf, int H,
                                                         int W, int rate, int IH, int IW, float* __restrict__ stats, int centre) {
    const int c = blockIdx.x, b = blockIdx.y, C = Cf + 3;
    float* st = stats + (size_t)b * STATS_STRIDE;
    if (c == 0 && threadIdx.x < STATS_VAR - C && C + (int)threadIdx.x < STATS_VAR) {
        st[C + threadIdx.x] = 0.0f; st[STATS_VAR + C + threadIdx.x] = 0.0f; st[STATS_OFF + C + threadIdx.x] = 0.0f;
        st[STATS_AMAX + C + threadIdx.x] = 0.0f; st[STATS_LAG + C + threadIdx.x] = 0.0f;
    }
    if (c == 0 && threadIdx.x < STATS_NFLAG) reinterpret_cast<int*>(st + STATS_FLAGS)[threadIdx.x] = 0;
    if (c >= STATS_VAR) return;
    const size_t bv = (size_t)b * (V + 1);
    // (the V source views and the reference view behind them: V + 1 views of the item)
    if (c < Cf) channel_stats<false>(feat + (bv * Cf + c) * H * W, (long long)Cf * H * W, V + 1, nullptr, H, W, 1, W, st + c, st + STATS_VAR + c, centre);
    else channel_stats<true>(rgb + (bv * 3 + (c - Cf)) * (size_t)IH * IW, 3ll * IH * IW, V + 1, nullptr, H, W, rate, IW, st + c, st + STATS_VAR + c, centre);
}

// the first block of a pack kernel: queue counters and slots cleared, the tiled kernel's guard set (header)
__device__ __forceinline__ void reset_queue_and_guard(int* __restrict__ queue, const float* __restrict__ stats, int B, int layout) {
    if (threadIdx.x < 64 && threadIdx.x != PICK_SLOT) queue[threadIdx.x] = threadIdx.x == LAYOUT_SLOT ? layout : 0;
    float off = 0.0f, var = 0.0f;
    for (int i = threadIdx.x; i < B * STATS_VAR; i += 256) {
        const int b = i / STATS_VAR, c = i - b * STATS_VAR;
        off += stats[(size_t)b * STATS_STRIDE + STATS_OFF + c];
        var += stats[(size_t)b * STATS_STRIDE + STATS_VAR + c];
    }
    __shared__ float scratch[4];
    off = block_sum_256(off, scratch);
    var = block_sum_256(var, scratch);
    if (threadIdx.x == 0) queue[NONCENTRED_SLOT] = off > 0.5f * var ? 1 : 0;
}

// Routing of the LDS-tiled kernel (L2; both entries -- the gather kernel reads the raw features from the NCHW tensor or from the float4
// copy alike): the conditioning measure of sweep_dist.hip ("Conditioning") per batch item, from
// the statistics of the pre-pass (the first 80 channels; scaled to C) and this call's candidates and sigma: flag 1 of the item's
// statistics row != 0 = the tiled kernel leaves the item's tiles to the gather kernel, which rounds like the reference.
__device__ __forceinline__ void route_ill_conditioned_items(float* __restrict__ stats, const SweepArgs& pa) {
    for (int b = threadIdx.x; b < pa.B; b += blockDim.x) {
        float* st = stats + (size_t)b * STATS_STRIDE;
        float sv = 0.0f, m2 = 0.0f;
        const int nc = pa.C < STATS_VAR ? pa.C : STATS_VAR;
        // (|mu|^2: the mean where the pre-pass centres, the squared offset it did NOT subtract where it does not)
        for (int c = 0; c < nc; ++c) { sv += st[STATS_VAR + c]; m2 += st[c] * st[c] + st[STATS_OFF + c]; }
        const float scale = (float)pa.C / (float)nc;
        float dhi = 0.0f, dlo = INFINITY;
        for (int k = 0; k < pa.D; ++k) { const float d = fabsf(pa.d_candi[k]); dhi = fmaxf(dhi, d); dlo = fminf(dlo, d); }
        const bool ill = pa.metric == 0 && (float)pa.V * scale * (2.0f * sv + m2) * (dhi - dlo) * 1.1920929e-7f > PDEPTH_COND_LIMIT_TILED * fabsf(pa.sigma);
        reinterpret_cast<int*>(st + STATS_FLAGS)[1] = ill ? 1 : 0;
    }
}

// NCHW -> channel-group-planar [C/4 + 2][H][W] float4: plane g < C/4 holds channels 4g .. 4g+3 of every texel minus mu
// (channels beyond C are zero), and the last two planes hold, for texel (x, y) and with s'(.) ABSENT (zero) outside the image:
//     plane C/4     : ( <s'(x,y),s'(x,y)>, <s'(x,y),s'(x+1,y)>, <s'(x,y),s'(x,y+1)>, <s'(x,y),s'(x+1,y+1)> + <s'(x+1,y),s'(x,y+1)> )
//     plane C/4 + 1 : ( <s'(x,y),mu>, 0, 0, 0 )
// (the two diagonal products only ever enter a cost as their sum; a consumer that adds the first component of the second
//  plane to it -- the kernels of earlier rounds, which found the second product there -- adds <s, 0> = 0 on the plain layout)
// One thread per texel, channels in order (sequential fma: deterministic); the neighbours' loads hit L1/L2.
template <bool CENTRE>   // (false: mu = 0 -- no statistics are read, the loop is the plain re-layout)
__global__ __launch_bounds__(256) void pack_c4_kernel(const float* __restrict__ src, long long bstride,
                                                      long long vstride, int V, int C, int H, int W,
                                                      float4* __restrict__ out, int* __restrict__ flags, int nflags, SweepArgs pa, int* queue,
                                                      const float* __restrict__ stats) {
    const int HW = H * W;
    // also clears the tile flags of this call (saves a memset launch)
    for (int i = (blockIdx.y * gridDim.x + blockIdx.x) * 256 + threadIdx.x; i < nflags; i += gridDim.x * gridDim.y * 256) flags[i] = 0;
    if (blockIdx.x == 0 && blockIdx.y == 0) {
        reset_queue_and_guard(queue, stats, pa.B, CENTRE ? LAYOUT_C4_CENTRED : LAYOUT_C4);
        if (threadIdx.x == 0) queue[PICK_SLOT] = 0;
        __syncthreads();
        if (pa.pick != 0) pick_for_launch(pa, queue, threadIdx.x, 256);
        if (pa.d_candi != nullptr) route_ill_conditioned_items(const_cast<float*>(stats), pa);   // (a sweep's pre-pass; pdepth_pack_source_f32 has no candidates)
    }
    // XCD-aware block order (workgroups are dealt round-robin over the 8 XCDs): every XCD packs one contiguous band
    // of rows, so the row below -- which another block of the same band loads as its own row -- hits that XCD's L2
    const int nb = gridDim.x, xcd = blockIdx.x & 7, qq = nb >> 3, rr = nb & 7;
    const int blk = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (blockIdx.x >> 3);
    const int pix = blk * 256 + threadIdx.x;
    const int bv = blockIdx.y;
    // the constants subtracted per channel: through LDS (a scalar load per channel in the loop below would sit in front of
    // every group of texel loads)
    __shared__ float mu[STATS_VAR];   // (CENTRE: C <= 72, every index of the loop below lies inside)
    if (CENTRE) {
        if (threadIdx.x < STATS_VAR) mu[threadIdx.x] = stats[(size_t)(bv / V) * STATS_STRIDE + threadIdx.x];
        __syncthreads();
    }
    if (pix >= HW) return;
    const int y = pix / W, x = pix - y * W;
    const bool hr = x + 1 < W, hd = y + 1 < H;
    const float* s = src + (size_t)(bv / V) * bstride + (size_t)(bv % V) * vstride + pix;
    const int ngrp = (C + 3) / 4;
    float4* o = out + (size_t)bv * (ngrp + 2) * HW + pix;
    pack_texel<CENTRE>(s, C, HW, W, hr, hd, mu, [&](int g, float4 q) { o[(size_t)g * HW] = q; });
}

// Encoder epilogue: what the host model does between its feature encoder and the sweep --
//     feats = cat(feat, avg_pool2d(rgb, rate))                 models/models.py:518-520, models/packnet.py:355-357
//     reference view = feats[:, -1], sources = feats[:, :-1]   models/models.py:530-534
// -- fused with the sweep's pre-pass: ONE pass over the encoder output writes the source views straight into the packed
// layout (as pack_c4_kernel) and the reference view as NCHW [B, Cf+3, H, W] (not centred: the sweep kernel centres the
// reference features it loads).  The concatenated [B, V+1, Cf+3, H, W] tensor is never materialised, and the sweep call
// that follows runs the packed entry.  feat [B*(V+1), Cf, H, W]; rgb [B*(V+1), 3, IH, IW] with IH >= H*rate, IW >= W*rate
// (the pooling ignores the remainder rows / columns, like avg_pool2d); view V of every item = reference.
// avg_pool2d as ATen computes it: window sum in row-major order, then divided by rate^2.
__global__ __launch_bounds__(256) void pack_views_kernel(const float* __restrict__ feat, const float* __restrict__ rgb, int V, int Cf,
                                                         int H, int W, int rate, int IH, int IW, float4* __restrict__ out,
                                                         float* __restrict__ ref_out, int* __restrict__ flags, int nflags, int* queue,
                                                         const float* __restrict__ stats, int B, int centred) {
    const int HW = H * W, C = Cf + 3;
    for (int i = (blockIdx.y * gridDim.x + blockIdx.x) * 256 + threadIdx.x; i < nflags; i += gridDim.x * gridDim.y * 256) flags[i] = 0;
    if (blockIdx.x == 0 && blockIdx.y == 0) {
        reset_queue_and_guard(queue, stats, B, centred ? LAYOUT_C4_CENTRED : LAYOUT_C4);
        if (threadIdx.x == 0) queue[PICK_SLOT] = 0;
    }
    const int nb = gridDim.x, xcd = blockIdx.x & 7, qq = nb >> 3, rr = nb & 7;   // XCD-aware block order, as pack_c4_kernel
    const int blk = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (blockIdx.x >> 3);
    const int pix = blk * 256 + threadIdx.x;
    const int bv = blockIdx.y, b = bv / (V + 1), v = bv % (V + 1);
    __shared__ float mu[STATS_VAR];
    if (threadIdx.x < STATS_VAR) mu[threadIdx.x] = stats[(size_t)b * STATS_STRIDE + threadIdx.x];
    __syncthreads();
    if (pix >= HW) return;
    const int y = pix / W, x = pix - y * W;
    const float* f = feat + (size_t)bv * Cf * HW + pix;
    const float* im = rgb + (size_t)bv * 3 * (size_t)IH * IW;
    auto pooled = [&](int c, int py, int px) -> float {   // avg_pool2d(rgb, rate)[c, py, px]
        const float* p = im + ((size_t)c * IH + (size_t)py * rate) * IW + (size_t)px * rate;
        float sum = 0.0f;
        if (rate == 4 && (IW & 3) == 0) {   // (the reference's quarter-resolution sweep: a window row is one 16-byte load; same order of the sum)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float4 q = *reinterpret_cast<const float4*>(p + (size_t)j * IW);
                sum += q.x; sum += q.y; sum += q.z; sum += q.w;
            }
            return sum / 16.0f;
        }
        for (int j = 0; j < rate; ++j)
            for (int i = 0; i < rate; ++i) sum += p[(size_t)j * IW + i];
        return sum / (float)(rate * rate);
    };
    if (v == V) {   // the reference view: NCHW copy + pooled image
        float* o = ref_out + (size_t)b * C * HW + pix;
        for (int c = 0; c < Cf; ++c) o[(size_t)c * HW] = f[(size_t)c * HW];
        for (int c = 0; c < 3; ++c) o[(size_t)(Cf + c) * HW] = pooled(c, y, x);
        return;
    }
    const bool hr = x + 1 < W, hd = y + 1 < H;
    const int ngrp = (C + 3) / 4;
    float4* o = out + (size_t)(b * V + v) * (ngrp + 2) * HW + pix;
    float n = 0.f, h = 0.f, vv = 0.f, d1 = 0.f, d2 = 0.f, mm = 0.f;
    const int i01 = hr ? 1 : 0, i10 = hd ? W : 0;
    auto accumulate = [&](float s00, float s01, float s10, float s11, float u) {
        n = __builtin_fmaf(s00, s00, n);
        h = __builtin_fmaf(s00, s01, h);
        vv = __builtin_fmaf(s00, s10, vv);
        d1 = __builtin_fmaf(s00, s11, d1);
        d2 = __builtin_fmaf(s01, s10, d2);
        mm = __builtin_fmaf(s00, u, mm);
    };
    // the groups of encoder channels: software-pipelined like pack_c4_kernel (16 loads of the next group in flight)
    const int gfeat = Cf / 4;   // groups that hold encoder channels only
    auto issue = [&](int g, float(&v_)[16]) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float* sc = f + (size_t)(g * 4 + j) * HW;
            v_[4 * j + 0] = sc[0]; v_[4 * j + 1] = sc[i01]; v_[4 * j + 2] = sc[i10]; v_[4 * j + 3] = sc[i01 + i10];
        }
    };
    auto finish = [&](int g, const float(&v_)[16]) {
        float c4[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float u = mu[min(g * 4 + j, STATS_VAR - 1)];   // (zero everywhere when the consumer does not centre)
            const float s00 = v_[4 * j + 0] - u;
            c4[j] = s00;
            accumulate(s00, hr ? v_[4 * j + 1] - u : 0.f, hd ? v_[4 * j + 2] - u : 0.f, hr && hd ? v_[4 * j + 3] - u : 0.f, u);
        }
        o[(size_t)g * HW] = make_float4(c4[0], c4[1], c4[2], c4[3]);
    };
    float va[16], vb[16];
    if (gfeat > 0) issue(0, va);
    for (int g = 0; g < gfeat; g += 2) {
        if (g + 1 < gfeat) issue(g + 1, vb);
        __builtin_amdgcn_sched_barrier(0);
        finish(g, va);
        if (g + 1 < gfeat) {
            if (g + 2 < gfeat) issue(g + 2, va);
            __builtin_amdgcn_sched_barrier(0);
            finish(g + 1, vb);
        }
    }
    // the groups that hold left-over encoder channels and the pooled image
    for (int g = gfeat; g < ngrp; ++g) {
        float c4[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = g * 4 + j;
            float s00 = 0.f, s01 = 0.f, s10 = 0.f, s11 = 0.f;
            const float u = mu[min(c, STATS_VAR - 1)];   // (zero beyond C)
            if (c < Cf) {
                const float* sc = f + (size_t)c * HW;
                s00 = sc[0] - u;
                s01 = hr ? sc[1] - u : 0.f;
                s10 = hd ? sc[W] - u : 0.f;
                s11 = hr && hd ? sc[W + 1] - u : 0.f;
            } else if (c < C) {
                s00 = pooled(c - Cf, y, x) - u;
                s01 = hr ? pooled(c - Cf, y, x + 1) - u : 0.f;
                s10 = hd ? pooled(c - Cf, y + 1, x) - u : 0.f;
                s11 = hr && hd ? pooled(c - Cf, y + 1, x + 1) - u : 0.f;
            }
            c4[j] = s00;
            accumulate(s00, s01, s10, s11, u);
        }
        o[(size_t)g * HW] = make_float4(c4[0], c4[1], c4[2], c4[3]);
    }
    o[(size_t)ngrp * HW] = make_float4(n, h, vv, d1 + d2);
    o[(size_t)(ngrp + 1) * HW] = make_float4(mm, 0.f, 0.f, 0.f);
}

// flag clear of a call on an already packed source when the kernel is chosen on the device (lab builds; else: a memset)
__global__ __launch_bounds__(256) void clear_and_pick_kernel(SweepArgs pa, int* flags, int nflags, int* queue) {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < nflags; i += gridDim.x * 256)
        if (flags + i != queue + PICK_SLOT && flags + i != queue + NONCENTRED_SLOT && flags + i != queue + LAYOUT_SLOT) flags[i] = 0;
    if (blockIdx.x == 0) {
        if (threadIdx.x == 0) queue[PICK_SLOT] = 0;
        __syncthreads();
        pick_for_launch(pa, queue, threadIdx.x, 256);
    }
}

// flag clear that keeps the guard slot the pre-pass wrote; the routing flags of the items for THIS call's candidates and sigma
__global__ __launch_bounds__(256) void clear_flags_kernel(int* flags, int nflags, int* queue, SweepArgs pa, float* stats) {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < nflags; i += gridDim.x * 256)
        if (flags + i != queue + NONCENTRED_SLOT && flags + i != queue + LAYOUT_SLOT) flags[i] = 0;
    if (blockIdx.x == 0 && pa.d_candi != nullptr) route_ill_conditioned_items(stats, pa);
}

size_t flag_only_bytes(int B, int H, int W) {
    const size_t tiles = (size_t)((W + TW - 1) / TW) * ((H + TH - 1) / TH);
    return ((size_t)B * tiles * sizeof(int) + 255) & ~(size_t)255;
}
size_t flag_bytes(int B, int H, int W) { return flag_only_bytes(B, H, W) + 256; }
// (the larger of the two staging layouts: channel-group-planar float4 + Gram planes | the distance-form kernel's fp16 planes, dist_layout.hpp)
size_t packed_bytes(int B, int V, int C, int H, int W) {
    const size_t c4 = (size_t)B * V * ((C + 3) / 4 + 2) * H * W * sizeof(float4);
    const size_t d16 = C <= dist::MAX_C ? (((size_t)B * V * (size_t)dist::view_bytes(C, H, W) + 255) & ~(size_t)255) : 0;
    return c4 > d16 ? c4 : d16;
}

}  // namespace

size_t sweep_ws_flag_only_bytes(int B, int H, int W) { return flag_only_bytes(B, H, W); }
size_t sweep_ws_flag_bytes(int B, int H, int W) { return flag_bytes(B, H, W); }
size_t sweep_ws_stats_offset(int B, int V, int C, int H, int W) {
    // flags + queue counters, packed source, then the list of tiles the fast cell-list kernel leaves to the generic one
    return flag_bytes(B, H, W) + packed_bytes(B, V, C, H, W) + flag_only_bytes(B, H, W);
}
size_t sweep_tiled_workspace_bytes(int B, int V, int C, int H, int W) {
    return sweep_ws_stats_offset(B, V, C, H, W) + (((size_t)B * STATS_STRIDE * sizeof(float) + 255) & ~(size_t)255);
}

// pre-pass of a call: channel statistics, packed source + Gram planes, tile flags and queue counters cleared
hipError_t launch_pack_c4(const SweepArgs& a, void* workspace, hipStream_t stream, bool centre) {
    int* flags = reinterpret_cast<int*>(workspace);
    float4* packed = reinterpret_cast<float4*>(static_cast<char*>(workspace) + flag_bytes(a.B, a.H, a.W));
    float* stats = reinterpret_cast<float*>(static_cast<char*>(workspace) + sweep_ws_stats_offset(a.B, a.V, a.C, a.H, a.W));
    const int HW = a.H * a.W;
    hipLaunchKernelGGL(feature_stats_kernel, dim3(a.C < STATS_VAR ? a.C : STATS_VAR, a.B), dim3(256), 0, stream, a.src, a.src_bstride, a.src_vstride, 1,
                       (const float*)nullptr, 0ll, a.C, a.H, a.W, stats, (centre && a.C <= 72) ? 1 : 0, (int*)nullptr, 0, (int*)nullptr, 0);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    dim3 pgrid((HW + 255) / 256, a.B * a.V);
    int* queue = reinterpret_cast<int*>(static_cast<char*>(workspace) + flag_only_bytes(a.B, a.H, a.W));
    if (centre && a.C <= 72)
        hipLaunchKernelGGL(pack_c4_kernel<true>, pgrid, dim3(256), 0, stream, a.src, a.src_bstride, a.src_vstride, a.V, a.C, a.H, a.W, packed,
                           flags, (int)(flag_only_bytes(a.B, a.H, a.W) / sizeof(int)), a, queue, stats);
    else
        hipLaunchKernelGGL(pack_c4_kernel<false>, pgrid, dim3(256), 0, stream, a.src, a.src_bstride, a.src_vstride, a.V, a.C, a.H, a.W, packed,
                           flags, (int)(flag_only_bytes(a.B, a.H, a.W) / sizeof(int)), a, queue, stats);
    return hipGetLastError();
}

hipError_t launch_feature_stats(const SweepArgs& a, float* stats, hipStream_t stream) {
    hipLaunchKernelGGL(feature_stats_kernel, dim3(a.C < STATS_VAR ? a.C : STATS_VAR, a.B), dim3(256), 0, stream, a.src, a.src_bstride, a.src_vstride, a.V,
                       a.ref, a.ref_bstride, a.C, a.H, a.W, stats, 1, (int*)nullptr, 0, (int*)nullptr, 0);
    return hipGetLastError();
}
hipError_t launch_view_stats(const SweepArgs& a, const float* feat, const float* rgb, int rate, int img_h, int img_w, float* stats, hipStream_t stream) {
    hipLaunchKernelGGL(view_stats_kernel, dim3(a.C < STATS_VAR ? a.C : STATS_VAR, a.B), dim3(256), 0, stream, feat, rgb, a.V, a.C - 3, a.H,
                       a.W, rate, img_h, img_w, stats, 1);
    return hipGetLastError();
}

// Pre-pass of a call whose sweep kernel packs the source itself (sweep_corr.hip): the channel statistics, and the workspace
// bookkeeping the pack kernel otherwise does.  The pack counters (two ints per batch item) live in the tile-list region.
bool sweep_ws_holds_pack_counters(int B, int H, int W) { return (size_t)2 * B * sizeof(int) <= flag_only_bytes(B, H, W); }
int* sweep_ws_pack_counters(const SweepArgs& a, void* workspace) {
    return reinterpret_cast<int*>(static_cast<char*>(workspace) + flag_bytes(a.B, a.H, a.W) + packed_bytes(a.B, a.V, a.C, a.H, a.W));
}
hipError_t launch_stats_only(const SweepArgs& a, void* workspace, hipStream_t stream) {
    float* stats = reinterpret_cast<float*>(static_cast<char*>(workspace) + sweep_ws_stats_offset(a.B, a.V, a.C, a.H, a.W));
    hipLaunchKernelGGL(feature_stats_kernel, dim3(a.C < STATS_VAR ? a.C : STATS_VAR, a.B), dim3(256), 0, stream, a.src, a.src_bstride, a.src_vstride, 1,
                       (const float*)nullptr, 0ll, a.C, a.H, a.W, stats, 1, reinterpret_cast<int*>(workspace), (int)(flag_bytes(a.B, a.H, a.W) / sizeof(int)),
                       sweep_ws_pack_counters(a, workspace), 2 * a.B);
    return hipGetLastError();
}

// the encoder epilogue (pack_views_kernel): a.C = Cf + 3, a.V source views, views V+1 per item in feat / rgb
hipError_t launch_pack_views(const SweepArgs& a, const float* feat, const float* rgb, int rate, int img_h, int img_w, float* ref_out,
                             void* workspace, hipStream_t stream, bool centre) {
    int* flags = reinterpret_cast<int*>(workspace);
    float4* packed = reinterpret_cast<float4*>(static_cast<char*>(workspace) + flag_bytes(a.B, a.H, a.W));
    float* stats = reinterpret_cast<float*>(static_cast<char*>(workspace) + sweep_ws_stats_offset(a.B, a.V, a.C, a.H, a.W));
    int* queue = reinterpret_cast<int*>(static_cast<char*>(workspace) + flag_only_bytes(a.B, a.H, a.W));
    const int HW = a.H * a.W;
    hipLaunchKernelGGL(view_stats_kernel, dim3(a.C < STATS_VAR ? a.C : STATS_VAR, a.B), dim3(256), 0, stream, feat, rgb, a.V, a.C - 3, a.H,
                       a.W, rate, img_h, img_w, stats, centre ? 1 : 0);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    dim3 pgrid((HW + 255) / 256, a.B * (a.V + 1));
    hipLaunchKernelGGL(pack_views_kernel, pgrid, dim3(256), 0, stream, feat, rgb, a.V, a.C - 3, a.H, a.W, rate, img_h, img_w, packed, ref_out,
                       flags, (int)(flag_only_bytes(a.B, a.H, a.W) / sizeof(int)), queue, stats, a.B, centre ? 1 : 0);
    return hipGetLastError();
}

// The pre-pass also clears the tile flags and queue counters; a call of the LDS-tiled kernel on an already packed source
// clears them itself (the guard slot the pre-pass wrote stays).
hipError_t clear_sweep_flags(const SweepArgs& a, void* workspace, hipStream_t stream) {
    int* queue = reinterpret_cast<int*>(static_cast<char*>(workspace) + flag_only_bytes(a.B, a.H, a.W));
    const int nflags = (int)(flag_bytes(a.B, a.H, a.W) / sizeof(int));
    if (a.pick != 0)
        hipLaunchKernelGGL(clear_and_pick_kernel, dim3((nflags + 2047) / 2048), dim3(256), 0, stream, a, reinterpret_cast<int*>(workspace), nflags, queue);
    else
        hipLaunchKernelGGL(clear_flags_kernel, dim3((nflags + 2047) / 2048), dim3(256), 0, stream, reinterpret_cast<int*>(workspace), nflags, queue, a,
                           reinterpret_cast<float*>(static_cast<char*>(workspace) + sweep_ws_stats_offset(a.B, a.V, a.C, a.H, a.W)));
    return hipGetLastError();
}

// 256-thread workgroups the device certainly holds at once: two per CU (every kernel of this library fits twice)
int sweep_resident_workgroups() { return 2 * sweep_device_cus(); }

// CU count of the current device (cached per device)
int sweep_device_cus() {
    static int cus[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (cus[dev] == 0) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        cus[dev] = n;
    }
    return cus[dev];
}

}  // namespace pdepth
