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

// Extended reduction (pdepth_dpv_reduce_ex_f32): the same single pass with optional extras, all from the registers
// that already hold the column --
//   addend   : x = logits + addend before the softmax   (feedback update log_softmax(BV_cur + BV_resi), models.py:694)
//   prob     : exp(logp), the decoder's input            (models.py:697 torch.exp(BV_cur_upd), :651)
//   variance : sum_k (d_k - E[d])^2 p_k                   (trainer/default_trainer.py:333-336)
//   quarter  : logp at every 4th row and column, [B,D,H/4,W/4] = F.interpolate(logp, scale_factor=0.25,
//              mode='nearest'), the next frame's prev_output (trainer/default_trainer.py:221)
struct DpvExtras {
    const float* addend;
    float* prob;
    float* variance;
    float* quarter;
    int W;
};

template <int RPL>
__global__ __launch_bounds__(256) void dpv_reduce_ex_vec4_kernel(const float* __restrict__ x, const float* __restrict__ dc,
                                                                 int D, int HW, float* logp, float* __restrict__ depth,
                                                                 DpvExtras ex) {
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int g = lane >> 4;
    const int quads = HW >> 2;
    const int q = wave * 16 + (lane & 15);
    const bool live = q < quads;
    const int b = blockIdx.y;
    const size_t off = (size_t)b * D * HW + (size_t)(live ? q : 0) * 4;
    float4 v[RPL];
    const float ninf = -INFINITY;
#pragma unroll
    for (int i = 0; i < RPL; ++i) {
        const int k = g + 4 * i;
        v[i] = make_float4(ninf, ninf, ninf, ninf);
        if (k < D && live) {
            v[i] = load_nt(x + off + (size_t)k * HW);
            if (ex.addend) {
                const float4 a = load_nt(ex.addend + off + (size_t)k * HW);
                v[i].x += a.x; v[i].y += a.y; v[i].z += a.z; v[i].w += a.w;
            }
        }
    }
    float4 m = v[0];
#pragma unroll
    for (int i = 1; i < RPL; ++i) {
        m.x = fmaxf(m.x, v[i].x); m.y = fmaxf(m.y, v[i].y); m.z = fmaxf(m.z, v[i].z); m.w = fmaxf(m.w, v[i].w);
    }
#pragma unroll
    for (int s = 16; s <= 32; s <<= 1) {
        const float4 o = shfl_xor4(m, s);
        m.x = fmaxf(m.x, o.x); m.y = fmaxf(m.y, o.y); m.z = fmaxf(m.z, o.z); m.w = fmaxf(m.w, o.w);
    }
    float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int i = 0; i < RPL; ++i) {
        if (g + 4 * i < D) {
            v[i].x -= m.x; v[i].y -= m.y; v[i].z -= m.z; v[i].w -= m.w;
            sum.x += expf(v[i].x); sum.y += expf(v[i].y); sum.z += expf(v[i].z); sum.w += expf(v[i].w);
        }
    }
#pragma unroll
    for (int s = 16; s <= 32; s <<= 1) {
        const float4 o = shfl_xor4(sum, s);
        sum.x += o.x; sum.y += o.y; sum.z += o.z; sum.w += o.w;
    }
    const float4 ls = make_float4(logf(sum.x), logf(sum.y), logf(sum.z), logf(sum.w));
    float4 e = make_float4(0.f, 0.f, 0.f, 0.f);
    // quarter-resolution copy: this lane's first pixel, when its row is a multiple of 4 (W % 4 == 0: a quad never
    // straddles rows and starts at a column that is a multiple of 4)
    const int pix = q * 4, py = pix / ex.W, pxq = (pix - py * ex.W) >> 2;
    const int Wq = ex.W >> 2, Hq = (HW / ex.W) >> 2;
    const bool qrow = ex.quarter && live && (py & 3) == 0 && (py >> 2) < Hq && pxq < Wq;
#pragma unroll
    for (int i = 0; i < RPL; ++i) {
        const int k = g + 4 * i;
        if (k < D) {
            const float4 lp = make_float4(v[i].x - ls.x, v[i].y - ls.y, v[i].z - ls.z, v[i].w - ls.w);
            const float4 p = make_float4(expf(lp.x), expf(lp.y), expf(lp.z), expf(lp.w));
            v[i] = p;   // (kept for the variance sweep)
            if (live) {
                if (logp) store_nt(logp + off + (size_t)k * HW, lp);
                if (ex.prob) store_nt(ex.prob + off + (size_t)k * HW, p);
                if (qrow) ex.quarter[((size_t)b * D + k) * Hq * Wq + (size_t)(py >> 2) * Wq + pxq] = lp.x;
            }
            const float dk = dc[k];
            e.x += dk * p.x; e.y += dk * p.y; e.z += dk * p.z; e.w += dk * p.w;
        }
    }
#pragma unroll
    for (int s = 16; s <= 32; s <<= 1) {
        const float4 o = shfl_xor4(e, s);
        e.x += o.x; e.y += o.y; e.z += o.z; e.w += o.w;
    }
    if (depth && live && g == 0) *reinterpret_cast<float4*>(depth + (size_t)b * HW + (size_t)q * 4) = e;
    if (ex.variance) {   // second sweep over the registers: sum_k (d_k - mean)^2 p_k with the mean just formed
        float4 var = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int i = 0; i < RPL; ++i) {
            const int k = g + 4 * i;
            if (k < D) {
                const float dk = dc[k];
                const float4 dd = make_float4(dk - e.x, dk - e.y, dk - e.z, dk - e.w);
                var.x += (dd.x * dd.x) * v[i].x; var.y += (dd.y * dd.y) * v[i].y;
                var.z += (dd.z * dd.z) * v[i].z; var.w += (dd.w * dd.w) * v[i].w;
            }
        }
#pragma unroll
        for (int s = 16; s <= 32; s <<= 1) {
            const float4 o = shfl_xor4(var, s);
            var.x += o.x; var.y += o.y; var.z += o.z; var.w += o.w;
        }
        if (live && g == 0) *reinterpret_cast<float4*>(ex.variance + (size_t)b * HW + (size_t)q * 4) = var;
    }
}

// any D / any H, W: one pixel per thread (re-reads hit L2)
__global__ __launch_bounds__(256) void dpv_reduce_ex_scalar_kernel(const float* x, const float* __restrict__ dc, int D,
                                                                   int HW, float* logp, float* __restrict__ depth,
                                                                   DpvExtras ex) {
    const int pix = blockIdx.x * 256 + threadIdx.x;
    if (pix >= HW) return;
    const int b = blockIdx.y;
    const size_t off = (size_t)b * D * HW + pix;
    auto at = [&](int k) { return x[off + (size_t)k * HW] + (ex.addend ? ex.addend[off + (size_t)k * HW] : 0.0f); };
    float m = -INFINITY;
    for (int k = 0; k < D; ++k) m = fmaxf(m, at(k));
    float s = 0.f;
    for (int k = 0; k < D; ++k) s += expf(at(k) - m);
    const float ls = logf(s);
    const int py = pix / ex.W, px = pix - py * ex.W, Wq = ex.W >> 2, Hq = (HW / ex.W) >> 2;
    const bool qpix = ex.quarter && (py & 3) == 0 && (px & 3) == 0 && (py >> 2) < Hq && (px >> 2) < Wq;
    float e = 0.f;
    for (int k = 0; k < D; ++k) {   // (logp may alias logits: nothing is read again after this sweep unless variance)
        const float lp = (at(k) - m) - ls;
        e += dc[k] * expf(lp);
    }
    float var = 0.f;
    for (int k = 0; k < D; ++k) {
        const float lp = (at(k) - m) - ls, p = expf(lp), dd = dc[k] - e;
        var += (dd * dd) * p;
        if (ex.prob) ex.prob[off + (size_t)k * HW] = p;
        if (qpix) ex.quarter[((size_t)b * D + k) * Hq * Wq + (size_t)(py >> 2) * Wq + (px >> 2)] = lp;
    }
    if (logp)
        for (int k = 0; k < D; ++k) logp[off + (size_t)k * HW] = (at(k) - m) - ls;   // last: logp may alias logits
    if (depth) depth[(size_t)b * HW + pix] = e;
    if (ex.variance) ex.variance[(size_t)b * HW + pix] = var;
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

hipError_t launch_dpv_reduce_ex(const float* logits, const float* addend, const float* d_candi, int B, int D, int H, int W,
                                float* logp, float* prob, float* depth, float* variance, float* quarter, hipStream_t stream) {
    const int HW = H * W;
    DpvExtras ex{addend, prob, variance, quarter, W};
    const bool vec = (W % 4 == 0) && aligned16(logits) && (!addend || aligned16(addend)) && (!logp || aligned16(logp)) &&
                     (!prob || aligned16(prob)) && (!depth || aligned16(depth)) && (!variance || aligned16(variance)) &&
                     D <= 128 && !(logp == logits && addend);   // (in place with an addend: the scalar kernel's order)
    if (vec) {
        dim3 grid((HW / 4 + 63) / 64, B);
        if (D <= 32)
            hipLaunchKernelGGL(dpv_reduce_ex_vec4_kernel<8>, grid, dim3(256), 0, stream, logits, d_candi, D, HW, logp, depth, ex);
        else if (D <= 64)
            hipLaunchKernelGGL(dpv_reduce_ex_vec4_kernel<16>, grid, dim3(256), 0, stream, logits, d_candi, D, HW, logp, depth, ex);
        else
            hipLaunchKernelGGL(dpv_reduce_ex_vec4_kernel<32>, grid, dim3(256), 0, stream, logits, d_candi, D, HW, logp, depth, ex);
    } else {
        dim3 grid((HW + 255) / 256, B);
        hipLaunchKernelGGL(dpv_reduce_ex_scalar_kernel, grid, dim3(256), 0, stream, logits, d_candi, D, HW, logp, depth, ex);
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
