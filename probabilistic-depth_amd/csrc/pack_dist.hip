// Pre-pass of the distance-form sweep kernel (sweep_dist.hip): the source views in the layout of dist_layout.hpp --
// centred, scaled by a power of two, split into fp16 high / low parts in matrix-operand order, with the squared norms
// and the squared neighbour differences the sweep needs per texel -- from the NCHW features (pdepth_sweep_dpv_f32,
// pdepth_pack_source_f32) or straight from the encoder output (pdepth_pack_views_f32: the reference's
// cat(feat, avg_pool2d(rgb)) of models/models.py:518-534 is never materialised).
//
// One thread per texel of the (H + 2) x (W + 2) image (ring of zero-feature texels = padding_mode 'zeros'), channels in
// groups of eight (= one 16-byte store of high parts and one of low parts), the next group's 32 loads in flight while a
// group is converted.  The kernel is a stream: 4 C bytes in, 16 * nplanes bytes out per texel.
#include <hip/hip_runtime.h>

#include "dist_layout.hpp"
#include "kernels.hpp"

namespace pdepth {

namespace {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));

// source accessors: raw feature c of texel (y, x) of view (b, v); the caller only asks for texels inside the image
struct NchwSource {
    const float* base;   // view (b, v)
    int HW, W;
    __device__ __forceinline__ float at(int c, int y, int x) const { return base[(size_t)c * HW + y * W + x]; }
};
// the encoder epilogue: channels < Cf from the encoder's feature maps, the rest = avg_pool2d(rgb, rate) as ATen computes it
// (window sum in row-major order, divided by rate^2)
struct ViewSource {
    const float* feat;   // [Cf, H, W] of the view
    const float* im;     // [3, IH, IW]
    int Cf, HW, W, rate, IH, IW;
    __device__ __forceinline__ float at(int c, int y, int x) const {
        if (c < Cf) return feat[(size_t)c * HW + y * W + x];
        const float* p = im + ((size_t)(c - Cf) * IH + (size_t)y * rate) * IW + (size_t)x * rate;
        float sum = 0.0f;
        if (rate == 4 && (IW & 3) == 0) {
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
    }
};

// mus[c] = mu[c] * 2^e (LDS, zeros beyond C), sc = 2^e.  out = the view's planes.  (xp, yp) = texel of the padded image.
template <int NCHK, typename Source>
__device__ __forceinline__ void pack_dist_texel(const Source& src, int C, int H, int W, int xp, int yp, const float* mus, float sc,
                                                char* __restrict__ out, int* __restrict__ item_flags) {
    const int x = xp - dist::RING, y = yp - dist::RING;
    const long long PB = dist::plane_bytes(H, W);
    const size_t toff = ((size_t)yp * dist::wp(W) + xp) * 16;
    // the texel and its right / lower / diagonal neighbours: inside the image?  (outside: the zero feature vector)
    const bool yin0 = (unsigned)y < (unsigned)H, yin1 = (unsigned)(y + 1) < (unsigned)H;
    const bool xin0 = (unsigned)x < (unsigned)W, xin1 = (unsigned)(x + 1) < (unsigned)W;
    const bool in[4] = {xin0 && yin0, xin1 && yin0, xin0 && yin1, xin1 && yin1};
    const int ya = min(max(y, 0), H - 1), yb = min(max(y + 1, 0), H - 1), xa = min(max(x, 0), W - 1), xb = min(max(x + 1, 0), W - 1);
    constexpr int NG = 4 * NCHK + 1;   // groups of 8 channels; the last one is the tail
    float n = 0.f, dx0 = 0.f, dy0 = 0.f, dd = 0.f, dx1 = 0.f;
    bool ovf = false;
    auto issue = [&](int g, float(&v)[32]) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = min(8 * g + j, C - 1);   // (channels beyond C: loaded from the last one, dropped below)
            v[4 * j + 0] = src.at(c, ya, xa); v[4 * j + 1] = src.at(c, ya, xb);
            v[4 * j + 2] = src.at(c, yb, xa); v[4 * j + 3] = src.at(c, yb, xb);
        }
    };
    auto finish = [&](int g, const float(&v)[32]) {
        h8 hh, ll;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = 8 * g + j;
            const bool has = c < C;   // uniform
            const float m = mus[min(c, dist::MAX_C + 7)];
            // x' = (x - mu) 2^e in one rounding (the scaling is exact); outside the image: x = 0
            const float s00 = __builtin_fmaf(has && in[0] ? v[4 * j + 0] : 0.f, sc, -m);
            const float s01 = __builtin_fmaf(has && in[1] ? v[4 * j + 1] : 0.f, sc, -m);
            const float s10 = __builtin_fmaf(has && in[2] ? v[4 * j + 2] : 0.f, sc, -m);
            const float s11 = __builtin_fmaf(has && in[3] ? v[4 * j + 3] : 0.f, sc, -m);
            ovf = ovf || !(fabsf(s00) <= 65000.0f);
            const float sx = fminf(fmaxf(s00, -65000.0f), 65000.0f);
            const _Float16 h = (_Float16)sx;
            hh[j] = h;
            ll[j] = (_Float16)(sx - (float)h);
            n = __builtin_fmaf(s00, s00, n);
            const float a = s00 - s01, b = s00 - s10, d1 = s00 - s11, d2 = s01 - s10, e = s10 - s11;
            dx0 = __builtin_fmaf(a, a, dx0);
            dy0 = __builtin_fmaf(b, b, dy0);
            dd = __builtin_fmaf(d1, d1, dd);
            dd = __builtin_fmaf(d2, d2, dd);
            dx1 = __builtin_fmaf(e, e, dx1);
        }
        if (g < 4 * NCHK) {
            *reinterpret_cast<h8*>(out + (size_t)g * PB + toff) = hh;
            *reinterpret_cast<h8*>(out + (size_t)(4 * NCHK + g) * PB + toff) = ll;
        } else {
            *reinterpret_cast<h8*>(out + (size_t)(8 * NCHK + 0) * PB + toff) = hh;
            *reinterpret_cast<h8*>(out + (size_t)(8 * NCHK + 1) * PB + toff) = ll;
            *reinterpret_cast<h8*>(out + (size_t)(8 * NCHK + 2) * PB + toff) = hh;
        }
    };
    float va[32], vb[32];
    issue(0, va);
#pragma unroll 1
    for (int g = 0; g < NG; g += 2) {
        if (g + 1 < NG) issue(g + 1, vb);
        __builtin_amdgcn_sched_barrier(0);
        finish(g, va);
        if (g + 1 < NG) {
            if (g + 2 < NG) issue(g + 2, va);
            __builtin_amdgcn_sched_barrier(0);
            finish(g + 1, vb);
        }
    }
    // specials: N as three fp16 pieces, and the constants that multiply the pixel's pieces of |r'|^2
    ovf = ovf || !(n < 2.0e9f);
    const dist::Pieces pn = dist::split_pieces(fminf(n, 2.0e9f));
    h8 sp;
    sp[0] = pn.p1; sp[1] = pn.p2; sp[2] = pn.p3;
    sp[3] = (_Float16)dist::PIECE_C1; sp[4] = (_Float16)dist::PIECE_C2; sp[5] = (_Float16)dist::PIECE_C3;
    sp[6] = (_Float16)0.f; sp[7] = (_Float16)0.f;
    *reinterpret_cast<h8*>(out + (size_t)(8 * NCHK + 3) * PB + toff) = sp;
    *reinterpret_cast<v4f*>(out + (size_t)(8 * NCHK + 4) * PB + toff) = v4f{dx0, dy0, dd, dx1};
    if (ovf) atomicOr(item_flags, 1);
}

// the scaled channel means of batch item b into LDS (mus: MAX_C + 8 floats), returns 2^e
__device__ __forceinline__ float load_item_scale(const float* __restrict__ stats, int b, float* mus, float* scratch) {
    const float* st = stats + (size_t)b * STATS_STRIDE;
    const int t = threadIdx.x;
    float am = t < STATS_VAR ? st[STATS_AMAX + t] : 0.0f;
    if (t < 64) {
        if (t + 64 < STATS_VAR) am = fmaxf(am, st[STATS_AMAX + t + 64]);
#pragma unroll
        for (int sh = 32; sh >= 1; sh >>= 1) am = fmaxf(am, __shfl_xor(am, sh));
        if (t == 0) scratch[0] = am;
    }
    __syncthreads();
    const float sc = ldexpf(1.0f, dist::scale_exponent(scratch[0]));
    if (t < dist::MAX_C + 8) mus[t] = st[t] * sc;   // (mu = 0 beyond C)
    __syncthreads();
    return sc;
}

__device__ __forceinline__ int xcd_block_order(int nb, int bx) {
    const int xcd = bx & 7, qq = nb >> 3, rr = nb & 7;
    return (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (bx >> 3);
}

// workspace bookkeeping every pack kernel does in its first block: queue counters and slots cleared, layout tag
__device__ __forceinline__ void reset_queue(int* __restrict__ queue) {
    if (threadIdx.x < 64) queue[threadIdx.x] = threadIdx.x == LAYOUT_SLOT ? LAYOUT_DIST16 : 0;
}

template <int NCHK>
__global__ __launch_bounds__(256) void pack_dist_kernel(const float* __restrict__ src, long long bstride, long long vstride, int V, int C,
                                                        int H, int W, char* __restrict__ out, int* __restrict__ flags, int nflags,
                                                        int* queue, float* __restrict__ stats) {
    for (int i = (blockIdx.y * gridDim.x + blockIdx.x) * 256 + threadIdx.x; i < nflags; i += gridDim.x * gridDim.y * 256) flags[i] = 0;
    if (blockIdx.x == 0 && blockIdx.y == 0) reset_queue(queue);
    const int bv = blockIdx.y, b = bv / V;
    __shared__ float mus[dist::MAX_C + 8];
    __shared__ float scratch[4];
    const float sc = load_item_scale(stats, b, mus, scratch);
    const int Wp = dist::wp(W), Hp = dist::hp(H);
    const int pix = xcd_block_order(gridDim.x, blockIdx.x) * 256 + threadIdx.x;
    if (pix >= Hp * Wp) return;
    const int yp = pix / Wp, xp = pix - yp * Wp;
    NchwSource s{src + (size_t)b * bstride + (size_t)(bv % V) * vstride, H * W, W};
    pack_dist_texel<NCHK>(s, C, H, W, xp, yp, mus, sc, out + (size_t)bv * dist::view_bytes(C, H, W),
                          reinterpret_cast<int*>(stats + (size_t)b * STATS_STRIDE + STATS_FLAGS));
}

// the encoder epilogue (sweep_pack.hip: pack_views_kernel says what it replaces): views 0..V-1 of an item into the packed
// layout, view V (the reference view) as NCHW [B, Cf + 3, H, W]
template <int NCHK>
__global__ __launch_bounds__(256) void pack_views_dist_kernel(const float* __restrict__ feat, const float* __restrict__ rgb, int V, int Cf,
                                                              int H, int W, int rate, int IH, int IW, char* __restrict__ out,
                                                              float* __restrict__ ref_out, int* __restrict__ flags, int nflags, int* queue,
                                                              float* __restrict__ stats) {
    const int HW = H * W, C = Cf + 3;
    for (int i = (blockIdx.y * gridDim.x + blockIdx.x) * 256 + threadIdx.x; i < nflags; i += gridDim.x * gridDim.y * 256) flags[i] = 0;
    if (blockIdx.x == 0 && blockIdx.y == 0) reset_queue(queue);
    const int bv = blockIdx.y, b = bv / (V + 1), v = bv % (V + 1);
    __shared__ float mus[dist::MAX_C + 8];
    __shared__ float scratch[4];
    const float sc = load_item_scale(stats, b, mus, scratch);
    const int pix = xcd_block_order(gridDim.x, blockIdx.x) * 256 + threadIdx.x;
    ViewSource s{feat + (size_t)bv * Cf * HW, rgb + (size_t)bv * 3 * (size_t)IH * IW, Cf, HW, W, rate, IH, IW};
    if (v == V) {   // the reference view: NCHW copy + pooled image
        if (pix >= HW) return;
        const int y = pix / W, x = pix - y * W;
        float* o = ref_out + (size_t)b * C * HW + pix;
        for (int c = 0; c < C; ++c) o[(size_t)c * HW] = s.at(c, y, x);
        return;
    }
    const int Wp = dist::wp(W), Hp = dist::hp(H);
    if (pix >= Hp * Wp) return;
    const int yp = pix / Wp, xp = pix - yp * Wp;
    pack_dist_texel<NCHK>(s, C, H, W, xp, yp, mus, sc, out + (size_t)(b * V + v) * dist::view_bytes(C, H, W),
                          reinterpret_cast<int*>(stats + (size_t)b * STATS_STRIDE + STATS_FLAGS));
}

}  // namespace

// statistics + packed source of the distance-form kernel (what launch_pack_c4 is for the other two)
hipError_t launch_pack_dist(const SweepArgs& a, void* workspace, hipStream_t stream) {
    int* flags = reinterpret_cast<int*>(workspace);
    char* packed = static_cast<char*>(workspace) + sweep_ws_flag_bytes(a.B, a.H, a.W);
    float* stats = reinterpret_cast<float*>(static_cast<char*>(workspace) + sweep_ws_stats_offset(a.B, a.V, a.C, a.H, a.W));
    int* queue = reinterpret_cast<int*>(static_cast<char*>(workspace) + sweep_ws_flag_only_bytes(a.B, a.H, a.W));
    hipError_t e = launch_feature_stats(a, stats, stream);
    if (e != hipSuccess) return e;
    const int npix = dist::hp(a.H) * dist::wp(a.W), nflags = (int)(sweep_ws_flag_only_bytes(a.B, a.H, a.W) / sizeof(int));
    dim3 grid((npix + 255) / 256, a.B * a.V);
#define PDEPTH_PACK_DIST(N) hipLaunchKernelGGL(pack_dist_kernel<N>, grid, dim3(256), 0, stream, a.src, a.src_bstride, a.src_vstride, a.V, a.C, a.H, \
                                               a.W, packed, flags, nflags, queue, stats)
    switch (dist::nchk(a.C)) {
        case 0: PDEPTH_PACK_DIST(0); break;
        case 1: PDEPTH_PACK_DIST(1); break;
        default: PDEPTH_PACK_DIST(2); break;
    }
#undef PDEPTH_PACK_DIST
    return hipGetLastError();
}

hipError_t launch_pack_views_dist(const SweepArgs& a, const float* feat, const float* rgb, int rate, int img_h, int img_w, float* ref_out,
                                  void* workspace, hipStream_t stream) {
    int* flags = reinterpret_cast<int*>(workspace);
    char* packed = static_cast<char*>(workspace) + sweep_ws_flag_bytes(a.B, a.H, a.W);
    float* stats = reinterpret_cast<float*>(static_cast<char*>(workspace) + sweep_ws_stats_offset(a.B, a.V, a.C, a.H, a.W));
    int* queue = reinterpret_cast<int*>(static_cast<char*>(workspace) + sweep_ws_flag_only_bytes(a.B, a.H, a.W));
    hipError_t e = launch_view_stats(a, feat, rgb, rate, img_h, img_w, stats, stream);
    if (e != hipSuccess) return e;
    const int npix = dist::hp(a.H) * dist::wp(a.W), nflags = (int)(sweep_ws_flag_only_bytes(a.B, a.H, a.W) / sizeof(int));
    dim3 grid((npix + 255) / 256, a.B * (a.V + 1));
#define PDEPTH_PACK_VIEWS_DIST(N) hipLaunchKernelGGL(pack_views_dist_kernel<N>, grid, dim3(256), 0, stream, feat, rgb, a.V, a.C - 3, a.H, a.W, rate, \
                                                     img_h, img_w, packed, ref_out, flags, nflags, queue, stats)
    switch (dist::nchk(a.C)) {
        case 0: PDEPTH_PACK_VIEWS_DIST(0); break;
        case 1: PDEPTH_PACK_VIEWS_DIST(1); break;
        default: PDEPTH_PACK_VIEWS_DIST(2); break;
    }
#undef PDEPTH_PACK_VIEWS_DIST
    return hipGetLastError();
}

}  // namespace pdepth
