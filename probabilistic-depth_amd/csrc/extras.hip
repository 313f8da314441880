// "Next" rows of SURVEY.md section 8(f): DPV Bayesian fusion and the FlowNet/PWC correlation op.
//
// dpv_fuse   -- upsample mode (models/models.py:666-672 with utils/img_utils.py:31-47, :360-375): build the
//               Gaussian soft label of a sparse depth map over the depth candidates, blend it with the
//               uniform DPV by the validity mask, multiply it into the network's DPV, renormalise, clamp,
//               take the log -- one kernel, one read of the log-DPV, two writes, instead of ~12 passes.
// correlation -- forward of the reference's only native operator (models/correlation_package/
//               correlation_cuda_kernel.cu:41-114; semantics pinned by models/correlation_native.py:13-23):
//               out[b, (dy+r)*(2r+1) + (dx+r), y, x] = mean_c x1[b,c,y,x] * x2[b,c,y+dy*s2,x+dx*s2] (zero padded).
//               It is dead code w.r.t. get_model (only PWCLite uses it), provided for completeness.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "geometry.hpp"
#include "kernels.hpp"

namespace pdepth {

// Single pass: lane = pixel, the D planes of the column live in registers between the three sweeps over k (Gaussian
// and its sum; fused product and its sum; normalise + write), so the log-DPV is read once and each output written
// once: 4*HW*(D*(1+outputs)+2) bytes.  DREG = planes held in registers (64 or 128); deeper volumes take the
// re-reading kernel below.
// exp / log / divide of the fusion kernels: hardware exp2 / log2 with an exact-argument reduction (exp_nonpos, geometry.hpp:
// ~1.5 ulp), log2 x ln 2 (~2 ulp, absolute 1e-7 near 1) and a refined reciprocal -- the libm forms cost ~20 instructions
// each, five per element, and made the kernel VALU bound (127 us for 402 MB).  -DPDEPTH_LIBM_FUSE restores them.
#ifdef PDEPTH_LIBM_FUSE
__device__ __forceinline__ float fuse_exp(float x) { return expf(x); }
__device__ __forceinline__ float fuse_log(float x) { return logf(x); }
__device__ __forceinline__ float fuse_div(float a, float b, float) { return a / b; }
__device__ __forceinline__ float fuse_rcp(float) { return 0.0f; }
#else
__device__ __forceinline__ float fuse_exp(float x) { return exp_nonpos(x); }
__device__ __forceinline__ float fuse_log(float x) { return __builtin_amdgcn_logf(x) * 0.693147180559945309417f; }
__device__ __forceinline__ float fuse_div(float a, float, float rb) { return a * rb; }
__device__ __forceinline__ float fuse_rcp(float b) { return refined_rcp(b); }
#endif

template <int DREG>
__global__ __launch_bounds__(256) void dpv_fuse_reg_kernel(const float* __restrict__ logp,
                                                           const float* __restrict__ dmaps,
                                                           const float* __restrict__ masks,
                                                           const float* __restrict__ dc, int D, int HW, float var,
                                                           float eps, float* __restrict__ fused,
                                                           float* __restrict__ logfused) {
    __shared__ float s_dc[DREG];
    for (int k = threadIdx.x; k < DREG; k += 256) s_dc[k] = k < D ? dc[k] : 0.0f;
    __syncthreads();
    const int pix = blockIdx.x * 256 + threadIdx.x;
    if (pix >= HW) return;
    const int b = blockIdx.y;
    const float dmap = dmaps[(size_t)b * HW + pix];
    const float mask = masks[(size_t)b * HW + pix];
    const float inv_mask = 1.0f - mask;
    const float sigma = sqrtf(var);
    const float two_var = 2.0f * (sigma * sigma);  // 2 * torch.pow(sig, 2)
    const float r_two_var = fuse_rcp(two_var);
    const float uni = 1.0f / (float)D;
    const float* lp = logp + (size_t)b * D * HW + pix;
    float v[DREG], x[DREG];
#pragma unroll
    for (int k = 0; k < DREG; ++k) x[k] = k < D ? __builtin_nontemporal_load(lp + (size_t)k * HW) : 0.0f;
    float sumg = 0.0f;
#pragma unroll
    for (int k = 0; k < DREG; ++k) {
        const float a = fabsf(s_dc[k] - dmap);
        v[k] = fuse_exp(fuse_div(-(a * a), two_var, r_two_var));
        if (k < D) sumg = sumg + v[k];
    }
    float sumf = 0.0f;
    const float r_sumg = fuse_rcp(sumg);
#pragma unroll
    for (int k = 0; k < DREG; ++k) {
        float t = fuse_div(v[k], sumg, r_sumg);
        if (t != t) t = -1.0f;                       // zero_invalid (img_utils.py:45)
        const float m = t * mask + uni * inv_mask;   // img_utils.py:371
        v[k] = fuse_exp(x[k] + fuse_log(fminf(fmaxf(m, eps), 1.0f)));
        if (k < D) sumf = sumf + v[k];
    }
    const float r_sumf = fuse_rcp(sumf);
    float* of = fused ? fused + (size_t)b * D * HW + pix : nullptr;
    float* ol = logfused ? logfused + (size_t)b * D * HW + pix : nullptr;
#pragma unroll
    for (int k = 0; k < DREG; ++k) {
        if (k < D) {
            const float f = fminf(fmaxf(fuse_div(v[k], sumf, r_sumf), eps), 1.0f);
            if (of) __builtin_nontemporal_store(f, of + (size_t)k * HW);
            if (ol) __builtin_nontemporal_store(fuse_log(f), ol + (size_t)k * HW);
        }
    }
}

// Any D: the column is re-read (from L2) in the second and third sweep.
__global__ __launch_bounds__(256) void dpv_fuse_kernel(const float* __restrict__ logp,
                                                       const float* __restrict__ dmaps,
                                                       const float* __restrict__ masks,
                                                       const float* __restrict__ dc, int D, int HW, float var,
                                                       float eps, float* __restrict__ fused,
                                                       float* __restrict__ logfused) {
    const int pix = blockIdx.x * 256 + threadIdx.x;
    if (pix >= HW) return;
    const int b = blockIdx.y;
    const float dmap = dmaps[(size_t)b * HW + pix];
    const float mask = masks[(size_t)b * HW + pix];
    const float inv_mask = 1.0f - mask;
    const float sigma = sqrtf(var);
    const float two_var = 2.0f * (sigma * sigma);  // 2 * torch.pow(sig, 2)
    const float uni = 1.0f / (float)D;
    const float* lp = logp + (size_t)b * D * HW + pix;
    float sumg = 0.0f;
    for (int k = 0; k < D; ++k) {
        const float a = fabsf(dc[k] - dmap);
        sumg = sumg + expf(-(a * a) / two_var);
    }
    auto tofuse = [&](int k) {
        const float a = fabsf(dc[k] - dmap);
        float t = expf(-(a * a) / two_var) / sumg;
        if (t != t) t = -1.0f;                       // zero_invalid (img_utils.py:45)
        const float m = t * mask + uni * inv_mask;   // img_utils.py:371
        return fminf(fmaxf(m, eps), 1.0f);            // clamp(eps, 1) keeps NaN out like torch.clamp
    };
    float sumf = 0.0f;
    for (int k = 0; k < D; ++k) sumf = sumf + expf(lp[(size_t)k * HW] + logf(tofuse(k)));
    float* of = fused ? fused + (size_t)b * D * HW + pix : nullptr;
    float* ol = logfused ? logfused + (size_t)b * D * HW + pix : nullptr;
    for (int k = 0; k < D; ++k) {
        float f = expf(lp[(size_t)k * HW] + logf(tofuse(k))) / sumf;
        f = fminf(fmaxf(f, eps), 1.0f);
        if (of) of[(size_t)k * HW] = f;
        if (ol) ol[(size_t)k * HW] = logf(f);
    }
}

// Mean and variance of the depth distribution per pixel (trainer/default_trainer.py:333-336):
//   z = exp(logDPV), mean = sum_k d_k z_k, variance = sum_k (d_k - mean)^2 z_k.
// Two sweeps over the column like the reference (the second one re-reads from L2).
__global__ __launch_bounds__(256) void dpv_moments_kernel(const float* __restrict__ dpv, const float* __restrict__ dc,
                                                          int D, int HW, int bv_log, float* __restrict__ mean,
                                                          float* __restrict__ var) {
    const int pix = blockIdx.x * 256 + threadIdx.x;
    if (pix >= HW) return;
    const int b = blockIdx.y;
    const float* x = dpv + (size_t)b * D * HW + pix;
    float m = 0.0f;
    for (int k = 0; k < D; ++k) {
        const float z = bv_log ? expf(x[(size_t)k * HW]) : x[(size_t)k * HW];
        m += dc[k] * z;
    }
    float s = 0.0f;
    for (int k = 0; k < D; ++k) {
        const float z = bv_log ? expf(x[(size_t)k * HW]) : x[(size_t)k * HW];
        const float e = dc[k] - m;
        s += (e * e) * z;
    }
    if (mean) mean[(size_t)b * HW + pix] = m;
    var[(size_t)b * HW + pix] = s;
}

// The same two sweeps with the column held in registers (D <= DREG): the volume is read once; same operations in the same
// order as the kernel above.
template <int DREG>
__global__ __launch_bounds__(256) void dpv_moments_reg_kernel(const float* __restrict__ dpv, const float* __restrict__ dc,
                                                              int D, int HW, int bv_log, float* __restrict__ mean,
                                                              float* __restrict__ var) {
    __shared__ float s_dc[DREG];
    for (int k = threadIdx.x; k < DREG; k += 256) s_dc[k] = k < D ? dc[k] : 0.0f;
    __syncthreads();
    const int pix = blockIdx.x * 256 + threadIdx.x;
    if (pix >= HW) return;
    const int b = blockIdx.y;
    const float* x = dpv + (size_t)b * D * HW + pix;
    float z[DREG];
#pragma unroll
    for (int k = 0; k < DREG; ++k) z[k] = k < D ? __builtin_nontemporal_load(x + (size_t)k * HW) : 0.0f;
    float m = 0.0f;
#pragma unroll
    for (int k = 0; k < DREG; ++k) {
        if (k < D) {
            if (bv_log) z[k] = expf(z[k]);
            m += s_dc[k] * z[k];
        }
    }
    float s = 0.0f;
#pragma unroll
    for (int k = 0; k < DREG; ++k) {
        if (k < D) {
            const float e = s_dc[k] - m;
            s += (e * e) * z[k];
        }
    }
    if (mean) mean[(size_t)b * HW + pix] = m;
    var[(size_t)b * HW + pix] = s;
}

hipError_t launch_dpv_moments(const float* dpv, const float* d_candi, int B, int D, int H, int W, int bv_log,
                              float* mean, float* var, hipStream_t stream) {
    dim3 grid((H * W + 255) / 256, B);
    if (D <= 64)
        hipLaunchKernelGGL(dpv_moments_reg_kernel<64>, grid, dim3(256), 0, stream, dpv, d_candi, D, H * W, bv_log, mean, var);
    else if (D <= 128)
        hipLaunchKernelGGL(dpv_moments_reg_kernel<128>, grid, dim3(256), 0, stream, dpv, d_candi, D, H * W, bv_log, mean, var);
    else
        hipLaunchKernelGGL(dpv_moments_kernel, grid, dim3(256), 0, stream, dpv, d_candi, D, H * W, bv_log, mean, var);
    return hipGetLastError();
}

hipError_t launch_dpv_fuse(const float* logp, const float* dmaps, const float* masks, const float* d_candi,
                           int B, int D, int H, int W, float var, float eps, float* fused, float* logfused,
                           hipStream_t stream) {
    dim3 grid((H * W + 255) / 256, B);
    if (D <= 64)
        hipLaunchKernelGGL(dpv_fuse_reg_kernel<64>, grid, dim3(256), 0, stream, logp, dmaps, masks, d_candi, D, H * W, var, eps, fused, logfused);
    else if (D <= 128)
        hipLaunchKernelGGL(dpv_fuse_reg_kernel<128>, grid, dim3(256), 0, stream, logp, dmaps, masks, d_candi, D, H * W, var, eps, fused, logfused);
    else
        hipLaunchKernelGGL(dpv_fuse_kernel, grid, dim3(256), 0, stream, logp, dmaps, masks, d_candi, D, H * W, var, eps, fused, logfused);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// correlation forward: block = 16x16 output pixels x ONE row of displacements (dy); the x2 tile of that row (halo in
// x only) and the x1 tile of a channel chunk are staged in LDS; every thread keeps the 2r+1 <= 9 sums of its pixel
// and row in registers.  Splitting the displacement rows over blocks gives (2r+1) x more workgroups (a 64x128 map has
// only 32 tiles) and 9 instead of 81 live accumulators; the price is that the x2 rows are staged once per row of
// displacements (from L2).  Sum over c in ascending order, one fma each, scaled by 1/C at the end.
// ---------------------------------------------------------------------------------------------------
constexpr int CT = 16;       // tile edge
constexpr int CCH = 8;       // channels per chunk
constexpr int RMAX = 4;      // max displacement radius (in units of stride2)

template <int R>
__global__ __launch_bounds__(256) void correlation_fwd_kernel(const float* __restrict__ x1,
                                                              const float* __restrict__ x2, int C, int H, int W,
                                                              int s2, float* __restrict__ out) {
    extern __shared__ float smem[];
    constexpr int ND = 2 * R + 1;
    const int halo = R * s2;
    const int TW = CT + 2 * halo;            // row length of the x2 tile
    float* t2 = smem;                        // [CCH][CT][TW]
    float* t1 = smem + CCH * CT * TW;        // [CCH][CT*CT]
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int x0 = blockIdx.x * CT, y0 = blockIdx.y * CT;
    const int i = blockIdx.z % ND, b = blockIdx.z / ND;
    const int dy = (i - R) * s2;
    const int x = x0 + tx, y = y0 + ty;
    const int HW = H * W;
    float acc[ND];
#pragma unroll
    for (int j = 0; j < ND; ++j) acc[j] = 0.0f;
    for (int c0 = 0; c0 < C; c0 += CCH) {
        __syncthreads();
        for (int idx = threadIdx.x; idx < CCH * CT * TW; idx += 256) {
            const int cc = idx / (CT * TW), rem = idx - cc * CT * TW;
            const int ry = rem / TW, rx = rem - ry * TW;
            const int gx = x0 - halo + rx, gy = y0 + dy + ry, c = c0 + cc;
            t2[idx] = (c < C && gx >= 0 && gx < W && gy >= 0 && gy < H) ? x2[((size_t)b * C + c) * HW + gy * W + gx] : 0.0f;
        }
#pragma unroll
        for (int cc = 0; cc < CCH; ++cc) {
            const int c = c0 + cc;
            t1[cc * 256 + threadIdx.x] = (c < C && x < W && y < H) ? x1[((size_t)b * C + c) * HW + y * W + x] : 0.0f;
        }
        __syncthreads();
#pragma unroll
        for (int cc = 0; cc < CCH; ++cc) {
            const float a = t1[cc * 256 + threadIdx.x];
            const float* row = t2 + (cc * CT + ty) * TW + tx;
#pragma unroll
            for (int j = 0; j < ND; ++j) acc[j] = __builtin_fmaf(a, row[j * s2], acc[j]);
        }
    }
    if (x < W && y < H) {
        const float inv = 1.0f / (float)C;
#pragma unroll
        for (int j = 0; j < ND; ++j) out[((size_t)b * ND * ND + i * ND + j) * HW + y * W + x] = acc[j] * inv;
    }
}

// ---------------------------------------------------------------------------------------------------
// correlation forward on the matrix pipe (C <= 256, max_displacement <= 16): the sum over the channels is a contraction,
// and for 16 neighbouring pixels of a row the texels of one displacement row they multiply with are 16 + 2 * max_displacement
// consecutive texels -- a banded 16 x (16 + 2 md) x C matrix product.  v_mfma_f32_16x16x4_f32 (exact fp32 products, fp32
// accumulation) computes it in ceil((16 + 2 md) / 16) blocks of 16 texels; of the 16 x 16 results of a block the (pixel,
// texel) pairs on the band's 2 r + 1 diagonals are outputs, the rest is matrix-pipe time nothing else wants.
//   wave  = 16 pixels of one image row, all displacement rows in turn (4 waves per workgroup: 4 pixel groups of the row);
//   B operand = x1[4 k + kq][y][x0 + n], held in registers for the whole wave (C / 4 of them);
//   A operand = x2[4 k + kq][y + dy][t0 + m] by buffer_load_dword (16 lanes = 64 contiguous bytes; outside the image:
//               out-of-range offset = 0 = the zero padding);
//   results pass through 9 x 16 floats of LDS per wave so that every displacement plane gets 64-byte row segments.
// ---------------------------------------------------------------------------------------------------
typedef float corr_v4f __attribute__((ext_vector_type(4)));
template <int KMAX>   // channel groups of 4 held in registers: C <= 4 * KMAX
__global__ __launch_bounds__(256) void correlation_fwd_mfma_kernel(const float* __restrict__ x1, const float* __restrict__ x2, int C, int H,
                                                                   int W, int r, int s2, float* __restrict__ out) {
    __shared__ float tr[4][9 * 16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n = lane & 15, kq = lane >> 4;
    const int HW = H * W, md = r * s2, ND = 2 * r + 1, nk = (C + 3) / 4, nblk = (16 + 2 * md + 15) / 16;
    const int xblocks = (W + 63) / 64;
    const int b = blockIdx.x / (H * xblocks), rem = blockIdx.x - b * (H * xblocks), y = rem / xblocks;
    const int x0 = (rem - y * xblocks) * 64 + wave * 16;
    if (x0 >= W) return;   // (wave-uniform; no barrier in this kernel)
    const int OOBO = 0x7fffffff;
    const __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc((void*)(x1 + (size_t)b * C * HW), 0, C * HW * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t r2 = __builtin_amdgcn_make_buffer_rsrc((void*)(x2 + (size_t)b * C * HW), 0, C * HW * 4, 0x00020000);
    float Rb[KMAX];
    {
        const bool okp = x0 + n < W;
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
            const int c = 4 * k + kq;
            Rb[k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r1, (k < nk && c < C && okp) ? (c * HW + y * W + x0 + n) * 4 : OOBO, 0, 0));
        }
    }
    const float inv = 1.0f / (float)C;
    float* const t = tr[wave];
    for (int di = 0; di < ND; ++di) {
        const int row2 = y + (di - r) * s2;
        const bool rowok = (unsigned)row2 < (unsigned)H;
        for (int blk = 0; blk < nblk; ++blk) {
            // four interleaved chains (pairwise at the end): a quarter of the additions per chain keeps the rounding of a
            // 256-channel sum inside the reference's own 1e-7 self-check bound, and four matrix instructions are independent
            corr_v4f accs[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) accs[q] = corr_v4f{0.f, 0.f, 0.f, 0.f};
            const int tx = x0 - md + 16 * blk + n;
            const bool okt = rowok && (unsigned)tx < (unsigned)W;
            const int base = (row2 * W + tx) * 4;
#pragma unroll
            for (int k = 0; k < KMAX; k += 4) {   // (fully unrolled: Rb stays in registers; groups beyond C are skipped uniformly)
                if (k < nk) {
                    float av[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int c = 4 * (k + q) + kq;
                        av[q] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r2, (okt && c < C) ? base + c * HW * 4 : OOBO, 0, 0));
                    }
#pragma unroll
                    for (int q = 0; q < 4; ++q) accs[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[q], Rb[k + q], accs[q], 0, 0, 0);
                }
            }
            const corr_v4f acc = (accs[0] + accs[1]) + (accs[2] + accs[3]);
            // lane (n, kq) holds texels m = 4 kq + i of the block for pixel n: offset of the texel from the pixel, in pixels
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int off = 16 * blk + 4 * kq + i - n - md;
                const int q = off / s2;   // (s2 >= 1; exact when the offset is one of the displacements)
                if (off >= -md && off <= md && q * s2 == off) t[(q + r) * 16 + n] = acc[i] * inv;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (wave-local: LDS operations of a wave complete in order)
        float* o = out + ((size_t)b * ND * ND + (size_t)di * ND) * HW + (size_t)y * W + x0;
        for (int idx = lane; idx < ND * 16; idx += 64) {
            const int dj = idx >> 4, px = idx & 15;
            if (x0 + px < W) o[(size_t)dj * HW + px] = t[idx];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
}

static bool correlation_forward_mfma_ok(int C, int H, int W, int radius, int stride2) {
    return C <= 256 && radius * stride2 <= 16 && (long long)C * H * W * 4 < (1ll << 31);
}

hipError_t launch_correlation_forward(const float* x1, const float* x2, int B, int C, int H, int W, int radius,
                                      int stride2, float* out, hipStream_t stream) {
#ifdef PDEPTH_LAB
    static const bool no_mfma = [] { const char* e = getenv("PDEPTH_CORR_NO_MFMA"); return e && e[0] == '1'; }();   // A/B timing
#else
    constexpr bool no_mfma = false;   // (the product library reads no environment)
#endif
    if (!no_mfma && radius >= 1 && radius <= RMAX && correlation_forward_mfma_ok(C, H, W, radius, stride2)) {
        const long long nwg = (long long)B * H * ((W + 63) / 64);
        if (nwg < (1ll << 31)) {
            if (C <= 64) hipLaunchKernelGGL((correlation_fwd_mfma_kernel<16>), dim3((unsigned)nwg), dim3(256), 0, stream, x1, x2, C, H, W, radius, stride2, out);
            else if (C <= 128) hipLaunchKernelGGL((correlation_fwd_mfma_kernel<32>), dim3((unsigned)nwg), dim3(256), 0, stream, x1, x2, C, H, W, radius, stride2, out);
            else hipLaunchKernelGGL((correlation_fwd_mfma_kernel<64>), dim3((unsigned)nwg), dim3(256), 0, stream, x1, x2, C, H, W, radius, stride2, out);
            return hipGetLastError();
        }
    }
    const int ND = 2 * radius + 1;
    dim3 grid((W + CT - 1) / CT, (H + CT - 1) / CT, B * ND);
    const int TW = CT + 2 * radius * stride2;
    const size_t lds = (size_t)(CCH * CT * TW + CCH * 256) * sizeof(float);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    auto go_launch = [&](auto kern) -> hipError_t {
        if (lds > 64 * 1024) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
        }
        hipLaunchKernelGGL(kern, grid, dim3(256), lds, stream, x1, x2, C, H, W, stride2, out);
        return hipGetLastError();
    };
    switch (radius) {
        case 1: return go_launch(correlation_fwd_kernel<1>);
        case 2: return go_launch(correlation_fwd_kernel<2>);
        case 3: return go_launch(correlation_fwd_kernel<3>);
        case 4: return go_launch(correlation_fwd_kernel<4>);
        default: return hipErrorInvalidValue;
    }
}

int correlation_max_radius() { return RMAX; }

// Backward of the correlation: with out[b,d,y,x] = (1/C) sum_c x1[b,c,y,x] * x2[b,c,y+dy,x+dx] (zero padded),
//   grad_x1[b,c,y,x] = (1/C) sum_d go[b,d,y,x]       * x2[b,c,y+dy,x+dx]
//   grad_x2[b,c,y,x] = (1/C) sum_d go[b,d,y-dy,x-dx] * x1[b,c,y-dy,x-dx]
// (correlation_cuda_kernel.cu:116-300 computes the same two sums over its padded NHWC repacks).
// Block = 16x16 pixels x CCH channels: the x1 and x2 halo tiles of the channel chunk are staged in LDS once and
// serve all (2r+1)^2 displacements; a thread reads the two gradient values of a displacement (its own pixel for
// grad_x1, the mirrored pixel for grad_x2) once from global memory and uses them for the CCH channels, so
// the gradient volume is read C/CCH times instead of C times and the features once per tile (+halo) instead of
// (2r+1)^2 times.  Per (pixel, channel) the sum runs over d in ascending order with one fma each, like the
// forward's twin in the oracle.
template <bool WANT1, bool WANT2>
__global__ __launch_bounds__(256) void correlation_bwd_kernel(const float* __restrict__ x1,
                                                              const float* __restrict__ x2,
                                                              const float* __restrict__ go, int C, int H, int W,
                                                              int r, int s2, float* __restrict__ g1,
                                                              float* __restrict__ g2) {
    extern __shared__ float smem[];
    const int halo = r * s2;
    const int TWH = CT + 2 * halo;
    float* t1 = smem;                       // [CCH][TWH][TWH] x1 with halo (for grad_x2)
    float* t2 = smem + CCH * TWH * TWH;     // [CCH][TWH][TWH] x2 with halo (for grad_x1)
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int x0 = blockIdx.x * CT, y0 = blockIdx.y * CT;
    const int c0 = blockIdx.z % ((C + CCH - 1) / CCH) * CCH, b = blockIdx.z / ((C + CCH - 1) / CCH);
    const int x = x0 + tx, y = y0 + ty;
    const int HW = H * W;
    const int nd = 2 * r + 1;
    for (int idx = threadIdx.x; idx < CCH * TWH * TWH; idx += 256) {
        const int cc = idx / (TWH * TWH), rem = idx - cc * TWH * TWH;
        const int ry = rem / TWH, rx = rem - ry * TWH;
        const int gx = x0 - halo + rx, gy = y0 - halo + ry, c = c0 + cc;
        const bool in = c < C && gx >= 0 && gx < W && gy >= 0 && gy < H;
        const size_t off = ((size_t)b * C + c) * HW + gy * W + gx;
        if (WANT2) t1[idx] = in ? x1[off] : 0.0f;
        if (WANT1) t2[idx] = in ? x2[off] : 0.0f;
    }
    __syncthreads();
    const bool inside = x < W && y < H;
    const float* gb = go + (size_t)b * nd * nd * HW;
    float a1[CCH], a2[CCH];
#pragma unroll
    for (int cc = 0; cc < CCH; ++cc) a1[cc] = a2[cc] = 0.0f;
    for (int i = 0; i < nd; ++i) {
        const int dy = (i - r) * s2;
        for (int j = 0; j < nd; ++j) {
            const int dx = (j - r) * s2;
            const float* gd = gb + (size_t)(i * nd + j) * HW;
            if (WANT1) {
                // x2 position paired with (y, x); outside the image the staged tile holds zeros, but the reference sum
                // skips those terms -- adding +-0 leaves every partial sum unchanged
                const float g = inside ? gd[y * W + x] : 0.0f;
                const float* p = t2 + (ty + halo + dy) * TWH + (tx + halo + dx);
#pragma unroll
                for (int cc = 0; cc < CCH; ++cc) a1[cc] = __builtin_fmaf(g, p[cc * TWH * TWH], a1[cc]);
            }
            if (WANT2) {
                const int ym = y - dy, xm = x - dx;  // x1 position whose pair is (y, x)
                const float g = (inside && ym >= 0 && ym < H && xm >= 0 && xm < W) ? gd[ym * W + xm] : 0.0f;
                const float* p = t1 + (ty + halo - dy) * TWH + (tx + halo - dx);
#pragma unroll
                for (int cc = 0; cc < CCH; ++cc) a2[cc] = __builtin_fmaf(g, p[cc * TWH * TWH], a2[cc]);
            }
        }
    }
    if (!inside) return;
    const float inv = 1.0f / (float)C;
#pragma unroll
    for (int cc = 0; cc < CCH; ++cc) {
        if (c0 + cc >= C) break;
        const size_t o = ((size_t)b * C + c0 + cc) * HW + y * W + x;
        if (WANT1) g1[o] = a1[cc] * inv;
        if (WANT2) g2[o] = a2[cc] * inv;
    }
}

hipError_t launch_correlation_backward(const float* x1, const float* x2, const float* go, int B, int C, int H, int W,
                                       int radius, int stride2, float* g1, float* g2, hipStream_t stream) {
    const int TWH = CT + 2 * radius * stride2;
    const size_t lds = (size_t)2 * CCH * TWH * TWH * sizeof(float);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    dim3 grid((W + CT - 1) / CT, (H + CT - 1) / CT, B * ((C + CCH - 1) / CCH));
    auto go_launch = [&](auto kern) -> hipError_t {
        if (lds > 64 * 1024) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
        }
        hipLaunchKernelGGL(kern, grid, dim3(256), lds, stream, x1, x2, go, C, H, W, radius, stride2, g1, g2);
        return hipGetLastError();
    };
    if (g1 && g2) return go_launch(correlation_bwd_kernel<true, true>);
    if (g1) return go_launch(correlation_bwd_kernel<true, false>);
    if (g2) return go_launch(correlation_bwd_kernel<false, true>);
    return hipSuccess;
}

// ---------------------------------------------------------------------------------------------------
// inverse_warp (utils/inverse_warp.py:174-210): depth-map driven warp used by the training losses
// (losses/loss_blocks.py:116 bilinear, :151 'nearest').  pixel2cam (:26-40), cam2pixel (:43-69, Z clamped at 1e-3,
// coordinates normalised with (w-1), (h-1)), then F.grid_sample with its default align_corners=False, zeros padding.
// Forward: one kernel.  Backward: the same per-pixel chain re-evaluated, the image gradient scattered with atomics
// (like ATen's grid_sampler backward) and the gradient with respect to the projected point (X, Y, Z) written per
// pixel; the 3x3 / 3x4 algebra behind it (depth, pose, intrinsics) is a handful of small matmuls on the host side.
// ---------------------------------------------------------------------------------------------------
namespace {

struct WarpSample {
    float X, Y, pz, Z;       // projected point, pz before the clamp
    float ix, iy;            // un-normalised sample position
    int x0, y0;              // top-left tap (clamped to [-2, size + 1])
    float nw, ne, sw, se;    // bilinear weights
    float w, e, n, s;
    bool finite, xi0, xi1, yi0, yi1;
    float xn, yn;
};

__device__ __forceinline__ WarpSample warp_sample(const float* __restrict__ ki, const float* __restrict__ pr, int x, int y,
                                                   float d, int H, int W) {
    WarpSample q;
    const float fx = (float)x, fy = (float)y;
    float cam[3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
        cam[i] = __builtin_fmaf(ki[i * 3 + 2], 1.0f, __builtin_fmaf(ki[i * 3 + 1], fy, ki[i * 3 + 0] * fx)) * d;
    float pc[3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
        pc[i] = __builtin_fmaf(pr[i * 4 + 2], cam[2], __builtin_fmaf(pr[i * 4 + 1], cam[1], pr[i * 4 + 0] * cam[0])) + pr[i * 4 + 3];
    q.X = pc[0]; q.Y = pc[1]; q.pz = pc[2];
    q.Z = fmaxf(pc[2], 1e-3f);
    q.xn = 2.0f * (pc[0] / q.Z) / (float)(W - 1) - 1.0f;
    q.yn = 2.0f * (pc[1] / q.Z) / (float)(H - 1) - 1.0f;
    q.ix = __builtin_fmaf(q.xn + 1.0f, (float)W / 2.0f, -0.5f);
    q.iy = __builtin_fmaf(q.yn + 1.0f, (float)H / 2.0f, -0.5f);
    const float xf = floorf(q.ix), yf = floorf(q.iy);
    q.w = q.ix - xf; q.e = 1.0f - q.w; q.n = q.iy - yf; q.s = 1.0f - q.n;
    q.nw = q.s * q.e; q.ne = q.s * q.w; q.sw = q.n * q.e; q.se = q.n * q.w;
    q.finite = (q.ix == q.ix) && (q.iy == q.iy);
    q.x0 = (int)fminf(fmaxf(xf, -2.0f), (float)(W + 1)); q.y0 = (int)fminf(fmaxf(yf, -2.0f), (float)(H + 1));
    q.xi0 = q.x0 >= 0 && q.x0 < W; q.xi1 = q.x0 + 1 >= 0 && q.x0 + 1 < W;
    q.yi0 = q.y0 >= 0 && q.y0 < H; q.yi1 = q.y0 + 1 >= 0 && q.y0 + 1 < H;
    return q;
}

// nearest tap of grid_sample(mode='nearest'): round half to even of the un-normalised position; -1 = outside / NaN
__device__ __forceinline__ int nearest_tap(const WarpSample& q, int H, int W) {
    if (!q.finite) return -1;
    const float rx = rintf(q.ix), ry = rintf(q.iy);
    if (!(rx >= 0.0f && rx < (float)W && ry >= 0.0f && ry < (float)H)) return -1;
    return (int)ry * W + (int)rx;
}

}  // namespace

template <int MODE>   // 0 bilinear, 1 nearest
__global__ __launch_bounds__(256) void inverse_warp_kernel(const float* __restrict__ img,
                                                           const float* __restrict__ depth,
                                                           const float* __restrict__ Kinv,
                                                           const float* __restrict__ proj, int C, int H, int W,
                                                           float* __restrict__ out, unsigned char* __restrict__ valid) {
    const int pix = blockIdx.x * 256 + threadIdx.x;
    const int HW = H * W;
    if (pix >= HW) return;
    const int b = blockIdx.y;
    const int y = pix / W, x = pix - y * W;
    const WarpSample q = warp_sample(Kinv + b * 9, proj + b * 12, x, y, depth[(size_t)b * HW + pix], H, W);
    if (valid) valid[(size_t)b * HW + pix] = (fmaxf(fabsf(q.xn), fabsf(q.yn)) <= 1.0f) ? 1 : 0;
    float* ob = out + (size_t)b * C * HW + pix;
    if (MODE == 1) {
        const int tap = nearest_tap(q, H, W);
        for (int c = 0; c < C; ++c) ob[(size_t)c * HW] = tap >= 0 ? img[((size_t)b * C + c) * HW + tap] : 0.0f;
        return;
    }
    const float* ib = img + (size_t)b * C * HW + (q.y0 * W + q.x0);
    for (int c = 0; c < C; ++c) {
        const float* p = ib + (size_t)c * HW;
        const float v00 = (q.finite && q.xi0 && q.yi0) ? p[0] : 0.0f;
        const float v01 = (q.finite && q.xi1 && q.yi0) ? p[1] : 0.0f;
        const float v10 = (q.finite && q.xi0 && q.yi1) ? p[W] : 0.0f;
        const float v11 = (q.finite && q.xi1 && q.yi1) ? p[W + 1] : 0.0f;
        ob[(size_t)c * HW] = __builtin_fmaf(v11, q.se, __builtin_fmaf(v10, q.sw, __builtin_fmaf(v01, q.ne, v00 * q.nw)));
    }
}

// grad_img [B,C,H,W] (zeroed by the launcher, atomics; may be NULL), grad_pc [B,3,H,W] = dL/d(X, Y, Z) of the
// projected point (zero in nearest mode, where the output does not depend on the position; may be NULL).
template <int MODE>
__global__ __launch_bounds__(256) void inverse_warp_bwd_kernel(const float* __restrict__ img,
                                                               const float* __restrict__ depth,
                                                               const float* __restrict__ Kinv,
                                                               const float* __restrict__ proj,
                                                               const float* __restrict__ go, int C, int H, int W,
                                                               float* __restrict__ grad_img, float* __restrict__ grad_pc) {
    const int pix = blockIdx.x * 256 + threadIdx.x;
    const int HW = H * W;
    if (pix >= HW) return;
    const int b = blockIdx.y;
    const int y = pix / W, x = pix - y * W;
    const WarpSample q = warp_sample(Kinv + b * 9, proj + b * 12, x, y, depth[(size_t)b * HW + pix], H, W);
    const float* gb = go + (size_t)b * C * HW + pix;
    float gix = 0.0f, giy = 0.0f;
    if (MODE == 1) {
        const int tap = nearest_tap(q, H, W);
        if (grad_img && tap >= 0)
            for (int c = 0; c < C; ++c) atomicAdd(grad_img + ((size_t)b * C + c) * HW + tap, gb[(size_t)c * HW]);
    } else {
        const size_t base = (size_t)b * C * HW + (q.y0 * W + q.x0);
        const bool t00 = q.finite && q.xi0 && q.yi0, t01 = q.finite && q.xi1 && q.yi0;
        const bool t10 = q.finite && q.xi0 && q.yi1, t11 = q.finite && q.xi1 && q.yi1;
        for (int c = 0; c < C; ++c) {
            const float g = gb[(size_t)c * HW];
            const float* p = img + base + (size_t)c * HW;
            const float v00 = t00 ? p[0] : 0.0f, v01 = t01 ? p[1] : 0.0f, v10 = t10 ? p[W] : 0.0f, v11 = t11 ? p[W + 1] : 0.0f;
            // d out / d ix = (v01 - v00) s + (v11 - v10) n,  d out / d iy = (v10 - v00) e + (v11 - v01) w
            gix = __builtin_fmaf(g, __builtin_fmaf(v11 - v10, q.n, (v01 - v00) * q.s), gix);
            giy = __builtin_fmaf(g, __builtin_fmaf(v11 - v01, q.w, (v10 - v00) * q.e), giy);
            if (grad_img) {
                float* gi = grad_img + base + (size_t)c * HW;
                if (t00) atomicAdd(gi, g * q.nw);
                if (t01) atomicAdd(gi + 1, g * q.ne);
                if (t10) atomicAdd(gi + W, g * q.sw);
                if (t11) atomicAdd(gi + W + 1, g * q.se);
            }
        }
    }
    if (grad_pc) {
        // ix = ((xn + 1) W - 1) / 2, xn = 2 (X / Z) / (W - 1) - 1, Z = max(pz, 1e-3) (clamp passes the gradient for pz >= 1e-3)
        float gX = 0.0f, gY = 0.0f, gZ = 0.0f;
        if (MODE == 0 && q.finite) {
            const float gxn = gix * ((float)W / 2.0f), gyn = giy * ((float)H / 2.0f);
            gX = gxn * 2.0f / ((float)(W - 1) * q.Z);
            gY = gyn * 2.0f / ((float)(H - 1) * q.Z);
            if (q.pz >= 1e-3f) gZ = -(gX * q.X + gY * q.Y) / q.Z;
        }
        float* o = grad_pc + (size_t)b * 3 * HW + pix;
        o[0] = gX; o[HW] = gY; o[2 * (size_t)HW] = gZ;
    }
}

hipError_t launch_inverse_warp(const float* img, const float* depth, const float* Kinv, const float* proj, int B,
                               int C, int H, int W, int mode, float* out, unsigned char* valid, hipStream_t stream) {
    dim3 grid((H * W + 255) / 256, B);
    if (mode == 0)
        hipLaunchKernelGGL(inverse_warp_kernel<0>, grid, dim3(256), 0, stream, img, depth, Kinv, proj, C, H, W, out, valid);
    else
        hipLaunchKernelGGL(inverse_warp_kernel<1>, grid, dim3(256), 0, stream, img, depth, Kinv, proj, C, H, W, out, valid);
    return hipGetLastError();
}

hipError_t launch_inverse_warp_backward(const float* img, const float* depth, const float* Kinv, const float* proj,
                                        const float* grad_out, int B, int C, int H, int W, int mode, float* grad_img,
                                        float* grad_pc, hipStream_t stream) {
    if (grad_img) {
        hipError_t e = hipMemsetAsync(grad_img, 0, (size_t)B * C * H * W * sizeof(float), stream);
        if (e != hipSuccess) return e;
    }
    dim3 grid((H * W + 255) / 256, B);
    if (mode == 0)
        hipLaunchKernelGGL(inverse_warp_bwd_kernel<0>, grid, dim3(256), 0, stream, img, depth, Kinv, proj, grad_out, C, H, W, grad_img, grad_pc);
    else
        hipLaunchKernelGGL(inverse_warp_bwd_kernel<1>, grid, dim3(256), 0, stream, img, depth, Kinv, proj, grad_out, C, H, W, grad_img, grad_pc);
    return hipGetLastError();
}

}  // namespace pdepth
