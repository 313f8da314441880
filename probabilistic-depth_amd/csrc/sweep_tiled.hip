// LDS-tiled plane sweep: the fast path of pdepth_sweep_{cost,dpv}_f32.
//
// Same arithmetic as sweep_direct.hip (reference op order, bit-faithful sample positions) but the
// bilinear taps come from LDS instead of global memory:
//
//   block   = 16x16 reference pixels (256 threads, 4 waves; a wave is 16 wide x 4 tall);
//   planes  = processed in groups of KP consecutive depth planes.  The per-plane geometry
//             (tap offset into the LDS window + 4 bilinear weights) of a group lives in registers,
//             so it is computed ONCE per (pixel, plane, view) -- not once per channel chunk;
//   window  = bounding box of every tap the block touches in the current plane group, staged
//             channel-chunk by channel-chunk (4*CG channels) into LDS as float4 texels
//             [g][row][col] with the row pitch padded to a multiple of 16 texels, which makes the
//             per-lane ds_read_b128 of a 16x4 wave bank-conflict free.  Texels outside the image
//             are staged as zeros, which IS padding_mode='zeros' -- no per-tap masks in the loop;
//   ref     = the block's reference features of the chunk, staged next to the window;
//   costs   = per-plane costs go to the output volume (cost, else logp, else workspace scratch),
//             accumulated over views in view order like homography.py:129; the fused epilogue
//             re-reads the pixel's D costs (L2-hot) for log_softmax + E[d].
//
// Consecutive planes of a group hit neighbouring texels, so a group's window is only a few texels
// wider than the tile; each HBM byte of the source map is read once into L2 and re-staged from
// there.  A block whose window does not fit NTEX_MAX texels (extreme poses) raises its tile flag
// and leaves the tile to the gather kernel of sweep_direct.hip -- results are identical.
#include <hip/hip_runtime.h>

#include "geometry.hpp"
#include "kernels.hpp"

namespace pdepth {

namespace {

constexpr int TILE = 16;          // tile edge (pixels)
constexpr int KP = 8;             // planes per group
constexpr int CG = 2;             // float4 channel groups per chunk (8 channels)
constexpr int NTEX_MAX = 1536;    // window texels per chunk group (LDS: NTEX_MAX*CG*16 B = 48 KB)

struct Bbox {
    int x0, y0, x1, y1;
};

__device__ __forceinline__ int wave_min(int v) {
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) v = min(v, __shfl_xor(v, s));
    return v;
}
__device__ __forceinline__ int wave_max(int v) {
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) v = max(v, __shfl_xor(v, s));
    return v;
}

}  // namespace

template <int METRIC>
__global__ __launch_bounds__(256, 2) void sweep_tiled_kernel(SweepArgs a, float* __restrict__ buf,
                                                             int* __restrict__ tile_flags, int tiles_x) {
    extern __shared__ __attribute__((aligned(16))) float4 lds4[];
    float4* win = lds4;                        // [CG][NTEX_MAX]
    float4* reft = lds4 + CG * NTEX_MAX;       // [CG][256]
    __shared__ int s_bbox[4];

    const int tid = threadIdx.x;
    const int lx = tid & 15, ly = tid >> 4;
    const int tile = blockIdx.x;
    const int b = blockIdx.y;
    const int tx0 = (tile % tiles_x) * TILE, ty0 = (tile / tiles_x) * TILE;
    const int x = tx0 + lx, y = ty0 + ly;
    const bool live = x < a.W && y < a.H;
    const int HW = a.H * a.W;
    const int p = live ? y * a.W + x : (min(y, a.H - 1) * a.W + min(x, a.W - 1));

    const float cx = a.cxcy[b * 2 + 0], cy = a.cxcy[b * 2 + 1];
    const float half_w = (float)a.W / 2.0f, half_h = (float)a.H / 2.0f;
    const float r0 = a.rays[((size_t)b * 3 + 0) * HW + p];
    const float r1 = a.rays[((size_t)b * 3 + 1) * HW + p];
    const float r2 = a.rays[((size_t)b * 3 + 2) * HW + p];
    const float* refb = a.ref + (size_t)b * a.ref_bstride;
    float* bufp = buf + (size_t)b * a.D * HW + p;
    const int nchunk = (a.C + 4 * CG - 1) / (4 * CG);

    for (int v = 0; v < a.V; ++v) {
        ViewXform xf;
        make_view_xform(a.K + b * 9, a.R + ((size_t)b * a.V + v) * 9, a.t + ((size_t)b * a.V + v) * 3,
                        a.blas_mode, xf);
        float t2a, t2b, t2c;
        ray_term2(xf, r0, r1, r2, t2a, t2b, t2c);
        const float* srcv = a.src + (size_t)b * a.src_bstride + (size_t)v * a.src_vstride;

        for (int k0 = 0; k0 < a.D; k0 += KP) {
            // ---- geometry of this plane group (registers) ---------------------------------
            int tx[KP], tyy[KP];
            float wnw[KP], wne[KP], wsw[KP], wse[KP];
            int bx0 = INT_MAX, by0 = INT_MAX, bx1 = INT_MIN, by1 = INT_MIN;
#pragma unroll
            for (int i = 0; i < KP; ++i) {
                const int k = min(k0 + i, a.D - 1);
                float ix, iy;
                plane_sample_pos(xf, t2a, t2b, t2c, a.d_candi[k], cx, cy, half_w, half_h, ix, iy);
                const Footprint f = make_footprint(ix, iy, a.W, a.H);
                tx[i] = f.x0; tyy[i] = f.y0;
                wnw[i] = f.nw; wne[i] = f.ne; wsw[i] = f.sw; wse[i] = f.se;
                if (f.mask != 0u && live) {  // at least one tap inside the image
                    bx0 = min(bx0, f.x0); bx1 = max(bx1, f.x0);
                    by0 = min(by0, f.y0); by1 = max(by1, f.y0);
                } else {
                    // all four taps read zero: point the sample at texel 0 of the window with
                    // weights that keep the reference's result (0 for finite positions, NaN for
                    // NaN positions because 0 * NaN = NaN as in ATen)
                    tx[i] = INT_MIN;
                }
            }
            // ---- block bounding box ---------------------------------------------------------
            bx0 = wave_min(bx0); by0 = wave_min(by0); bx1 = wave_max(bx1); by1 = wave_max(by1);
            __syncthreads();  // previous group's window reads and s_bbox reads are done
            if (tid == 0) { s_bbox[0] = INT_MAX; s_bbox[1] = INT_MAX; s_bbox[2] = INT_MIN; s_bbox[3] = INT_MIN; }
            __syncthreads();
            if ((tid & 63) == 0) {
                atomicMin(&s_bbox[0], bx0); atomicMin(&s_bbox[1], by0);
                atomicMax(&s_bbox[2], bx1); atomicMax(&s_bbox[3], by1);
            }
            __syncthreads();
            Bbox w{s_bbox[0], s_bbox[1], s_bbox[2], s_bbox[3]};
            const bool empty = w.x0 > w.x1;  // every sample of the group is fully out of bounds
            if (empty) { w.x0 = 0; w.x1 = 0; w.y0 = 0; w.y1 = 0; }
            const int WC = ((w.x1 - w.x0 + 2) + 15) & ~15;  // +1 east tap, pitch multiple of 16
            const int WR = w.y1 - w.y0 + 2;                  // +1 south tap
            const int ntex = WC * WR;
            if (ntex > NTEX_MAX) {  // block-uniform: leave the tile to the gather kernel
                if (tid == 0) tile_flags[b * gridDim.x + tile] = 1;
                return;
            }
            int off[KP];
#pragma unroll
            for (int i = 0; i < KP; ++i) {
                const bool oob = tx[i] == INT_MIN;
                off[i] = oob ? 0 : (tyy[i] - w.y0) * WC + (tx[i] - w.x0);
                if (oob) {  // keep NaN weights (NaN position), zero finite ones
                    wnw[i] = wnw[i] * 0.0f; wne[i] = wne[i] * 0.0f; wsw[i] = wsw[i] * 0.0f; wse[i] = wse[i] * 0.0f;
                }
            }
            float acc[KP];
#pragma unroll
            for (int i = 0; i < KP; ++i) acc[i] = 0.0f;

            // ---- channel chunks ---------------------------------------------------------------
            for (int ch = 0; ch < nchunk; ++ch) {
                if (ch > 0) __syncthreads();  // readers of the previous chunk are done
                const int cbase = ch * 4 * CG;
                // stage the window (zeros outside the image / beyond C); WC is a multiple of 16, so
                // the 16x16 thread grid walks it without integer division
                const int c0 = cbase;
#pragma unroll
                for (int g = 0; g < CG; ++g) {
                    const int c = c0 + 4 * g;
                    for (int row = ly; row < WR; row += TILE) {
                        const int gy = w.y0 + row;
                        const bool yin = gy >= 0 && gy < a.H && !empty;
                        for (int col = lx; col < WC; col += TILE) {
                            const int gx = w.x0 + col;
                            float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
                            if (yin && gx >= 0 && gx < a.W) {
                                const float* s = srcv + (size_t)c * HW + gy * a.W + gx;
                                val.x = (c + 0 < a.C) ? s[0] : 0.f;
                                val.y = (c + 1 < a.C) ? s[(size_t)HW] : 0.f;
                                val.z = (c + 2 < a.C) ? s[2 * (size_t)HW] : 0.f;
                                val.w = (c + 3 < a.C) ? s[3 * (size_t)HW] : 0.f;
                            }
                            win[g * NTEX_MAX + row * WC + col] = val;
                        }
                    }
                }
                // stage the reference features of the tile
#pragma unroll
                for (int g = 0; g < CG; ++g) {
                    const int c = cbase + 4 * g;
                    const float* r = refb + (size_t)c * HW + p;
                    float4 val;
                    val.x = (c + 0 < a.C) ? r[0] : 0.f;
                    val.y = (c + 1 < a.C) ? r[(size_t)HW] : 0.f;
                    val.z = (c + 2 < a.C) ? r[2 * (size_t)HW] : 0.f;
                    val.w = (c + 3 < a.C) ? r[3 * (size_t)HW] : 0.f;
                    reft[g * 256 + tid] = val;
                }
                __syncthreads();
#pragma unroll
                for (int g = 0; g < CG; ++g) {
                    const float4 rf = reft[g * 256 + tid];
                    const float4* wg = win + g * NTEX_MAX;
#pragma unroll
                    for (int i = 0; i < KP; ++i) {
                        const float4 s00 = wg[off[i]];
                        const float4 s01 = wg[off[i] + 1];
                        const float4 s10 = wg[off[i] + WC];
                        const float4 s11 = wg[off[i] + WC + 1];
#define PDEPTH_TAP(comp)                                                            \
    {                                                                               \
        float val = s00.comp * wnw[i];                                              \
        val = __builtin_fmaf(s01.comp, wne[i], val);                                \
        val = __builtin_fmaf(s10.comp, wsw[i], val);                                \
        val = __builtin_fmaf(s11.comp, wse[i], val);                                \
        const float diff = val - rf.comp;                                           \
        acc[i] = acc[i] + (METRIC == 0 ? diff * diff : fabsf(diff));                \
    }
                        PDEPTH_TAP(x) PDEPTH_TAP(y) PDEPTH_TAP(z) PDEPTH_TAP(w)
#undef PDEPTH_TAP
                    }
                }
            }
            // channels beyond C were staged as zeros on both sides: they add (0-0)^2 = 0, except for
            // NaN-weight samples where they add NaN -- which the reference produces as well.
            if (live) {
#pragma unroll
                for (int i = 0; i < KP; ++i) {
                    if (k0 + i < a.D) {
                        float* o = bufp + (size_t)(k0 + i) * HW;
                        const float c = acc[i] / a.sigma;
                        *o = (v == 0) ? (0.0f + c) : (*o + c);
                    }
                }
            }
        }
    }

    // ---- epilogue: log-softmax over D + expectation from the pixel's own costs ---------------
    if (!live) return;
    if (a.cost_out && a.cost_out != buf) {
        float* o = a.cost_out + (size_t)b * a.D * HW + p;
        for (int k = 0; k < a.D; ++k) o[(size_t)k * HW] = bufp[(size_t)k * HW];
    }
    if (a.logp_out || a.depth_out) {
        float m = -INFINITY;
        for (int k = 0; k < a.D; ++k) m = fmaxf(m, bufp[(size_t)k * HW]);
        float s = 0.0f;
        for (int k = 0; k < a.D; ++k) s = s + expf(bufp[(size_t)k * HW] - m);
        const float ls = logf(s);
        float e = 0.0f;
        float* o = a.logp_out ? a.logp_out + (size_t)b * a.D * HW + p : nullptr;
        for (int k = 0; k < a.D; ++k) {
            const float lp = (bufp[(size_t)k * HW] - m) - ls;  // read before the (aliasing) store
            if (o) o[(size_t)k * HW] = lp;
            e = e + a.d_candi[k] * expf(lp);
        }
        if (a.depth_out) a.depth_out[(size_t)b * HW + p] = e;
    }
}

size_t sweep_tiled_workspace_bytes(int B, int D, int H, int W, bool need_scratch) {
    const size_t tiles = (size_t)((W + TILE - 1) / TILE) * ((H + TILE - 1) / TILE);
    size_t bytes = ((size_t)B * tiles * sizeof(int) + 255) & ~(size_t)255;
    if (need_scratch) bytes += (size_t)B * D * H * W * sizeof(float);
    return bytes;
}

// Launches the tiled kernel, then the gather kernel on the tiles it flagged.
hipError_t launch_sweep_tiled(const SweepArgs& a, void* workspace, hipStream_t stream) {
    const int tiles_x = (a.W + TILE - 1) / TILE, tiles_y = (a.H + TILE - 1) / TILE;
    const int tiles = tiles_x * tiles_y;
    int* flags = reinterpret_cast<int*>(workspace);
    const size_t flag_bytes = ((size_t)a.B * tiles * sizeof(int) + 255) & ~(size_t)255;
    float* buf = a.logp_out ? a.logp_out : a.cost_out;
    if (!buf) buf = reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) + flag_bytes);
    hipError_t e = hipMemsetAsync(flags, 0, (size_t)a.B * tiles * sizeof(int), stream);
    if (e != hipSuccess) return e;
    const size_t lds = (size_t)(CG * NTEX_MAX + CG * 256) * sizeof(float4);
    dim3 grid(tiles, a.B);
    if (a.metric == 0)
        hipLaunchKernelGGL(sweep_tiled_kernel<0>, grid, dim3(256), lds, stream, a, buf, flags, tiles_x);
    else
        hipLaunchKernelGGL(sweep_tiled_kernel<1>, grid, dim3(256), lds, stream, a, buf, flags, tiles_x);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    return launch_sweep_direct_flagged(a, flags, tiles_x, tiles, stream);
}

}  // namespace pdepth
