// Channel statistics of the feature views (the constant that is subtracted per channel, the fp16 scale, the guard's energy and
// lagged spread): the body of feature_stats_kernel / view_stats_kernel (sweep_pack.hip), shared with the pack kernel of the
// distance-form layout, which runs it in its first workgroups (pack_dist.hip: one pre-pass launch instead of two).
#pragma once
#include <hip/hip_runtime.h>

#include "kernels.hpp"

namespace pdepth {
namespace stats_body {

constexpr int STATS_ROWS = 8;

__device__ __forceinline__ float block_sum_256(float v, float* scratch) {
#pragma unroll
    for (int sh = 32; sh >= 1; sh >>= 1) v = v + __shfl_xor(v, sh);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = v;
    __syncthreads();
    return (scratch[0] + scratch[1]) + (scratch[2] + scratch[3]);
}

// rows sampled for the statistics: STATS_ROWS rows spread evenly (all rows of a small image)
__device__ __forceinline__ int stats_row(int i, int nrows, int H) { return min(H - 1, ((2 * i + 1) * H) / (2 * nrows)); }

// Block (c, b): mean and variance of channel c of item b over the sampled rows.  POOLED: the channel is avg_pool2d(rgb, rate)
// of the encoder epilogue (pack_views_kernel), computed on the fly like there.  The samples of a thread are independent
// loads issued together (32 at a time), then summed: the kernel is a few microseconds of latency, not a dependent chain.
__device__ __forceinline__ float block_max_256(float v, float* scratch) {
#pragma unroll
    for (int sh = 32; sh >= 1; sh >>= 1) v = fmaxf(v, __shfl_xor(v, sh));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = v;
    __syncthreads();
    return fmaxf(fmaxf(scratch[0], scratch[1]), fmaxf(scratch[2], scratch[3]));
}

// The sample: rows of EVERY view that enters the cost -- the V source views and, where the caller has it (the NCHW entry, the
// encoder epilogue), the reference view (`extra`): the maximum that sets the fp16 scale, the energy and the lagged spread of the
// guard and the constant that is subtracted are the pooled ones.  (Round 5 sampled source view 0 only: a view unlike it, or a
// reference with another exposure, was invisible to the scale and to the guard.)  nv views `vstride` floats apart.
template <bool POOLED>
__device__ __forceinline__ void channel_stats(const float* __restrict__ plane, long long vstride, int nv, const float* __restrict__ extra, int H,
                                              int W, int rate, int IW, float* __restrict__ mu_out, float* __restrict__ var_out, int centre) {
    __shared__ float scratch[4];
    const int nvw = nv + (extra ? 1 : 0);
    // (STATS_ROWS rows in all, at least two per view: the kernel is a round of loads and four block reductions -- latency, 8 us)
    const int nrows = min(H, max(2, STATS_ROWS / nvw)), per_view = nrows * W, total = nvw * per_view;
    // the partner of a sample for the lagged spread: STATS_LAG_PX texels to the right (to the left in the last columns)
    const int lag = W > 2 * STATS_LAG_PX ? STATS_LAG_PX : (W > 1 ? W / 2 : 0);
    float s = 0.0f, s2 = 0.0f, am = 0.0f, dl = 0.0f;
    constexpr int NU = 16;   // samples of a thread in flight together (and as many partners)
    for (int i0 = threadIdx.x; i0 < total; i0 += 256 * NU) {
        float v[NU], w[NU];
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int i = i0 + 256 * u;
            v[u] = 0.0f; w[u] = 0.0f;
            if (i < total) {
                const int vw = i / per_view, iv = i - vw * per_view;
                const float* pl = vw < nv ? plane + (size_t)vw * vstride : extra;
                const int r = iv / W, x = iv - r * W, y = stats_row(r, nrows, H);
                const int x2 = x + lag < W ? x + lag : x - lag;
                if (POOLED) {
                    // (the mean of the pooled channel = the mean of the image itself: one image row per sampled map row, every
                    //  rate-th column -- the variance, which only feeds the guards, is the image's, an upper bound)
                    v[u] = pl[((size_t)y * rate) * IW + (size_t)x * rate];
                    w[u] = pl[((size_t)y * rate) * IW + (size_t)x2 * rate];
                } else {
                    v[u] = pl[(size_t)y * W + x];
                    w[u] = pl[(size_t)y * W + x2];
                }
            }
        }
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            s += v[u]; s2 = __builtin_fmaf(v[u], v[u], s2);
            am = fmaxf(am, fabsf(v[u]));
            const float d = v[u] - w[u];
            dl = __builtin_fmaf(d, d, dl);
        }
    }
    const float cnt = (float)total;
    const float mean = block_sum_256(s, scratch) / cnt;
    const float msq = block_sum_256(s2, scratch) / cnt;
    const float amax = block_max_256(am, scratch);
    const float dlag = block_sum_256(dl, scratch) / cnt;
    if (threadIdx.x == 0) {
        // (non-finite features: no centring -- the statistics would poison every pixel of the item)
        const bool fin = fabsf(mean) < 1.0e30f && msq < 1.0e30f;
        *mu_out = (centre && fin) ? mean : 0.0f;
        // spread around the constant that is subtracted, and the offset that is NOT removed (for the tiled kernel's guard)
        var_out[0] = fin ? fmaxf(msq - mean * mean, 0.0f) : 0.0f;
        var_out[STATS_OFF - STATS_VAR] = (centre || !fin) ? 0.0f : mean * mean;
        // (NaN / inf features: amax as it comes out -- the distance-form kernels then take the scale 1)
        var_out[STATS_AMAX - STATS_VAR] = amax + (centre && fin ? fabsf(mean) : 0.0f);
        var_out[STATS_LAG - STATS_VAR] = fin && lag > 0 ? 0.5f * dlag : var_out[0];
    }
}


}  // namespace stats_body
}  // namespace pdepth
