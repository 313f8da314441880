// DPV reduction kernels: log-softmax over the depth axis + expectation, one pass over HBM.
//
// Replaces F.log_softmax(x, dim=1) (models/models.py:560,637,694,:351; packnet.py:394)
// followed by dpv_to_depthmap (utils/img_utils.py:52-61; called per item at
// trainer/default_trainer.py:229-233).  The reference makes 5 passes over the D x H x W
// volume (log_softmax read+write, exp, mul, sum); here the logits are read once, held in
// registers, and logp + depth are written once: 4*HW*(2D+1) bytes per item.
//
// Layout of a wave (vec4 kernel): lane = (plane group g = lane>>4, pixel quad q = lane&15).
// A quad is 4 consecutive pixels (one 16-byte load); the 16 quads of a wave cover 256
// contiguous bytes of every plane row, and the 4 plane groups interleave the D planes
// (k = g + 4*i).  The per-pixel max / sum-exp / sum d*exp are combined across the 4 plane
// groups with two xor-shuffles each (lanes l, l^16, l^32, l^48).
#include <hip/hip_runtime.h>

#include "kernels.hpp"

namespace pdepth {

typedef float v4f __attribute__((ext_vector_type(4)));

// streaming (non-temporal) 16-byte accesses: the volume is read once and written once (+5 % measured)
__device__ __forceinline__ float4 load_nt(const float* p) {
    const v4f t = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(p));
    return make_float4(t.x, t.y, t.z, t.w);
}
__device__ __forceinline__ void store_nt(float* p, float4 v) {
    __builtin_nontemporal_store(v4f{v.x, v.y, v.z, v.w}, reinterpret_cast<v4f*>(p));
}

__device__ __forceinline__ float4 shfl_xor4(float4 v, int m) {
    return make_float4(__shfl_xor(v.x, m), __shfl_xor(v.y, m), __shfl_xor(v.z, m),
                       __shfl_xor(v.w, m));
}

template <int RPL>
__global__ __launch_bounds__(256) void dpv_reduce_vec4_kernel(const float* __restrict__ x,
                                                              const float* __restrict__ dc,
                                                              int D, int HW, float* logp,
                                                              float* __restrict__ depth) {
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int g = lane >> 4;
    const int quads = HW >> 2;
    const int q = wave * 16 + (lane & 15);
    const bool live = q < quads;
    const int b = blockIdx.y;
    const float* xb = x + (size_t)b * D * HW + (size_t)(live ? q : 0) * 4;

    float4 v[RPL];
    const float ninf = -INFINITY;
#pragma unroll
    for (int i = 0; i < RPL; ++i) {
        const int k = g + 4 * i;
        v[i] = (k < D && live) ? load_nt(xb + (size_t)k * HW)
                               : make_float4(ninf, ninf, ninf, ninf);
    }
    float4 m = v[0];
#pragma unroll
    for (int i = 1; i < RPL; ++i) {
        m.x = fmaxf(m.x, v[i].x); m.y = fmaxf(m.y, v[i].y);
        m.z = fmaxf(m.z, v[i].z); m.w = fmaxf(m.w, v[i].w);
    }
#pragma unroll
    for (int s = 16; s <= 32; s <<= 1) {
        const float4 o = shfl_xor4(m, s);
        m.x = fmaxf(m.x, o.x); m.y = fmaxf(m.y, o.y); m.z = fmaxf(m.z, o.z); m.w = fmaxf(m.w, o.w);
    }
    float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int i = 0; i < RPL; ++i) {
        const int k = g + 4 * i;
        if (k < D) {
            v[i].x -= m.x; v[i].y -= m.y; v[i].z -= m.z; v[i].w -= m.w;
            sum.x += expf(v[i].x); sum.y += expf(v[i].y);
            sum.z += expf(v[i].z); sum.w += expf(v[i].w);
        }
    }
#pragma unroll
    for (int s = 16; s <= 32; s <<= 1) {
        const float4 o = shfl_xor4(sum, s);
        sum.x += o.x; sum.y += o.y; sum.z += o.z; sum.w += o.w;
    }
    const float4 ls = make_float4(logf(sum.x), logf(sum.y), logf(sum.z), logf(sum.w));
    float4 e = make_float4(0.f, 0.f, 0.f, 0.f);
    float* lb = logp ? logp + (size_t)b * D * HW + (size_t)(live ? q : 0) * 4 : nullptr;
#pragma unroll
    for (int i = 0; i < RPL; ++i) {
        const int k = g + 4 * i;
        if (k < D) {
            const float4 lp = make_float4(v[i].x - ls.x, v[i].y - ls.y, v[i].z - ls.z, v[i].w - ls.w);
            if (lb && live) store_nt(lb + (size_t)k * HW, lp);
            const float dk = dc[k];
            e.x += dk * expf(lp.x); e.y += dk * expf(lp.y);
            e.z += dk * expf(lp.z); e.w += dk * expf(lp.w);
        }
    }
#pragma unroll
    for (int s = 16; s <= 32; s <<= 1) {
        const float4 o = shfl_xor4(e, s);
        e.x += o.x; e.y += o.y; e.z += o.z; e.w += o.w;
    }
    if (depth && live && g == 0)
        *reinterpret_cast<float4*>(depth + (size_t)b * HW + (size_t)q * 4) = e;
}

// Any D / any HW: one pixel per thread, three sweeps over the column (re-reads hit L2).
__global__ __launch_bounds__(256) void dpv_reduce_scalar_kernel(const float* x,
                                                                const float* __restrict__ dc,
                                                                int D, int HW, float* logp,
                                                                float* __restrict__ depth) {
    const int pix = blockIdx.x * 256 + threadIdx.x;
    if (pix >= HW) return;
    const int b = blockIdx.y;
    const float* xb = x + (size_t)b * D * HW + pix;
    float m = -INFINITY;
    for (int k = 0; k < D; ++k) m = fmaxf(m, xb[(size_t)k * HW]);
    float s = 0.f;
    for (int k = 0; k < D; ++k) s += expf(xb[(size_t)k * HW] - m);
    const float ls = logf(s);
    float e = 0.f;
    float* lb = logp ? logp + (size_t)b * D * HW + pix : nullptr;
    for (int k = 0; k < D; ++k) {
        const float lp = (xb[(size_t)k * HW] - m) - ls;  // read before the aliasing store
        if (lb) lb[(size_t)k * HW] = lp;
        e += dc[k] * expf(lp);
    }
    if (depth) depth[(size_t)b * HW + pix] = e;
}

// Expectation with the wave layout of dpv_reduce_vec4_kernel: lane = (plane group g, pixel quad q), all loads of a
// lane issued up front (RPL 16-byte non-temporal loads in flight per lane), partial sums combined by xor-shuffles.
template <bool BV_LOG, int RPL>
__global__ __launch_bounds__(256) void dpv_expect_vec4_kernel(const float* __restrict__ x,
                                                              const float* __restrict__ dc, int D, int HW,
                                                              float* __restrict__ depth) {
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int g = lane >> 4;
    const int quads = HW >> 2;
    const int q = wave * 16 + (lane & 15);
    const bool live = q < quads;
    const int b = blockIdx.y;
    const float* xb = x + (size_t)b * D * HW + (size_t)(live ? q : 0) * 4;
    float4 v[RPL];
#pragma unroll
    for (int i = 0; i < RPL; ++i) {
        const int k = g + 4 * i;
        v[i] = (k < D && live) ? load_nt(xb + (size_t)k * HW) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    float4 e = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int i = 0; i < RPL; ++i) {
        const int k = g + 4 * i;
        if (k < D) {
            const float dk = dc[k];
            e.x += dk * (BV_LOG ? expf(v[i].x) : v[i].x); e.y += dk * (BV_LOG ? expf(v[i].y) : v[i].y);
            e.z += dk * (BV_LOG ? expf(v[i].z) : v[i].z); e.w += dk * (BV_LOG ? expf(v[i].w) : v[i].w);
        }
    }
#pragma unroll
    for (int s = 16; s <= 32; s <<= 1) {
        const float4 o = shfl_xor4(e, s);
        e.x += o.x; e.y += o.y; e.z += o.z; e.w += o.w;
    }
    if (live && g == 0) *reinterpret_cast<float4*>(depth + (size_t)b * HW + (size_t)q * 4) = e;
}

template <bool BV_LOG, int VEC>
__global__ __launch_bounds__(256) void dpv_expect_kernel(const float* __restrict__ x,
                                                         const float* __restrict__ dc, int D,
                                                         int HW, float* __restrict__ depth) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int n = HW / VEC;
    if (i >= n) return;
    const int b = blockIdx.y;
    const float* xb = x + (size_t)b * D * HW + (size_t)i * VEC;
    float e[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) e[j] = 0.f;
    for (int k = 0; k < D; ++k) {
        const float dk = dc[k];
        float v[VEC];
        if (VEC == 4) {
            const float4 t = *reinterpret_cast<const float4*>(xb + (size_t)k * HW);
            v[0] = t.x; v[1 % VEC] = t.y; v[2 % VEC] = t.z; v[3 % VEC] = t.w;
        } else {
            v[0] = xb[(size_t)k * HW];
        }
#pragma unroll
        for (int j = 0; j < VEC; ++j) e[j] += dk * (BV_LOG ? expf(v[j]) : v[j]);
    }
#pragma unroll
    for (int j = 0; j < VEC; ++j) depth[(size_t)b * HW + (size_t)i * VEC + j] = e[j];
}

static bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

hipError_t launch_dpv_reduce(const float* logits, const float* d_candi, int B, int D, int H,
                             int W, float* logp, float* depth, hipStream_t stream) {
    const int HW = H * W;
    const bool vec = (HW % 4 == 0) && aligned16(logits) && (!logp || aligned16(logp)) &&
                     (!depth || aligned16(depth)) && D <= 128;
    if (vec) {
        const int quads = HW / 4;
        dim3 grid((quads + 63) / 64, B);
        if (D <= 32)
            hipLaunchKernelGGL(dpv_reduce_vec4_kernel<8>, grid, dim3(256), 0, stream, logits, d_candi, D, HW, logp, depth);
        else if (D <= 64)
            hipLaunchKernelGGL(dpv_reduce_vec4_kernel<16>, grid, dim3(256), 0, stream, logits, d_candi, D, HW, logp, depth);
        else
            hipLaunchKernelGGL(dpv_reduce_vec4_kernel<32>, grid, dim3(256), 0, stream, logits, d_candi, D, HW, logp, depth);
    } else {
        dim3 grid((HW + 255) / 256, B);
        hipLaunchKernelGGL(dpv_reduce_scalar_kernel, grid, dim3(256), 0, stream, logits, d_candi, D, HW, logp, depth);
    }
    return hipGetLastError();
}

hipError_t launch_dpv_expect(const float* dpv, const float* d_candi, int B, int D, int H, int W,
                             int bv_log, float* depth, hipStream_t stream) {
    const int HW = H * W;
    const bool vec = (HW % 4 == 0) && aligned16(dpv) && aligned16(depth);
    if (vec && D <= 128) {
        dim3 grid((HW / 4 + 63) / 64, B);
#define PDEPTH_EXPECT(RPL)                                                                                              \
    if (bv_log) hipLaunchKernelGGL((dpv_expect_vec4_kernel<true, RPL>), grid, dim3(256), 0, stream, dpv, d_candi, D, HW, depth); \
    else hipLaunchKernelGGL((dpv_expect_vec4_kernel<false, RPL>), grid, dim3(256), 0, stream, dpv, d_candi, D, HW, depth);
        if (D <= 32) { PDEPTH_EXPECT(8) } else if (D <= 64) { PDEPTH_EXPECT(16) } else { PDEPTH_EXPECT(32) }
#undef PDEPTH_EXPECT
    } else if (vec) {
        dim3 grid((HW / 4 + 255) / 256, B);
        if (bv_log) hipLaunchKernelGGL((dpv_expect_kernel<true, 4>), grid, dim3(256), 0, stream, dpv, d_candi, D, HW, depth);
        else hipLaunchKernelGGL((dpv_expect_kernel<false, 4>), grid, dim3(256), 0, stream, dpv, d_candi, D, HW, depth);
    } else {
        dim3 grid((HW + 255) / 256, B);
        if (bv_log) hipLaunchKernelGGL((dpv_expect_kernel<true, 1>), grid, dim3(256), 0, stream, dpv, d_candi, D, HW, depth);
        else hipLaunchKernelGGL((dpv_expect_kernel<false, 1>), grid, dim3(256), 0, stream, dpv, d_candi, D, HW, depth);
    }
    return hipGetLastError();
}

}  // namespace pdepth
