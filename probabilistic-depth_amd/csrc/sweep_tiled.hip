// LDS-tiled plane sweep: the fast path of pdepth_sweep_{cost,dpv}_f32.
//
// Same bit-faithful sample positions and bilinear weights as sweep_direct.hip, but the taps come from
// LDS instead of global memory and the per-channel arithmetic is folded into five fma-class ops.
//
//   block   = 16x4 reference pixels x 4 plane groups = 256 threads.  Wave w of the block owns the
//             same 64 pixels (lane = 16 wide x 4 tall) and plane group w: KP = 8 consecutive depth
//             planes of the current "super group" of 4*KP = 32 planes.  The per-plane geometry of a
//             thread (tap offset into the LDS window + 4 bilinear weights, 8 planes) lives in
//             registers and is computed ONCE per (pixel, plane, view);
//   window  = bounding box of every tap the block touches in the current super group, staged four
//             channels at a time into LDS as float4 texels [row][col] (double buffered: chunk ch+1
//             travels global -> registers while chunk ch is computed from LDS; one barrier per
//             chunk).  The row pitch is a multiple of 16 texels, which makes the per-lane
//             ds_read_b128 of a 16x4 wave bank-conflict free.  Texels outside the image are staged
//             as zeros, which IS padding_mode='zeros' -- no per-tap masks in the inner loop;
//   ref     = the tile's reference features of the chunk, staged next to the window;
//   sum     = five VALU ops per (pixel, plane, channel): the reference feature enters the bilinear fma
//             chain as its initial addend (diff = fma(s00,nw,-r) ... fma(s11,se,.)) and the square is
//             accumulated with one fma.  The reference rounds the interpolated value and the square
//             separately (7 ops); the difference is ~1e-7 relative per term, two orders below the
//             parity tolerance (the gather kernel of sweep_direct.hip keeps the reference's op order);
//   costs   = cost[k][pixel] of the tile in LDS (D x 64 floats), accumulated over views in view
//             order like homography.py:129; the fused epilogue (log_softmax over D + E[d]) runs on
//             those with the 4 waves splitting the planes, so nothing but the requested outputs is
//             ever written to HBM.
//
// Consecutive planes hit neighbouring texels, so a super group's window is only a few texels larger
// than the tile and each source texel is staged from L2 a handful of times per chunk instead of being
// gathered 4 x D times.  A block whose window does not fit (extreme poses) raises its tile flag and
// leaves the tile to the gather kernel of sweep_direct.hip; a window that is merely too large is first
// split into 2 or 4 parts of consecutive planes.  Both kernels agree to rounding (~1e-7 relative per term).
#include <hip/hip_runtime.h>

#include "geometry.hpp"
#include "kernels.hpp"

namespace pdepth {

namespace {

constexpr int TW = 16, TH = 4;    // tile (pixels); one wave covers it
constexpr int NPG = 4;            // plane groups per block (= waves)
constexpr int KP = 8;             // planes per group
constexpr int SG = NPG * KP;      // planes per super group
constexpr int NBUF = 2;           // LDS window buffers
constexpr int NTEX_MAX = 1024;    // window texels per buffer (LDS: NTEX_MAX*NBUF*16 B = 32 KB)
constexpr int SLOTS = 4;          // sub-blocks of a window (256 texels each, register-staged prefetch)

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
__global__ __launch_bounds__(256, 3) void sweep_tiled_kernel(SweepArgs a, int* __restrict__ tile_flags,
                                                             int tiles_x) {
    extern __shared__ __attribute__((aligned(16))) float4 lds4[];
    float4* win = lds4;                                   // [NBUF][NTEX_MAX]
    float4* reft = lds4 + NBUF * NTEX_MAX;                // [NBUF][64]
    float* costs = reinterpret_cast<float*>(reft + NBUF * 64);  // [D][64]
    float* red = costs + (size_t)a.D * 64;                // [NPG][64]
    __shared__ int s_bbox[4];

    const int tid = threadIdx.x;
    const int pgl = tid >> 6;        // plane group of this wave
    const int lane = tid & 63;       // pixel of the tile
    const int lx = lane & 15, ly = lane >> 4;
    // XCD-aware tile order: workgroups are dealt round-robin over the 8 XCDs, so blocks i and i+8
    // share an L2.  Give every XCD one contiguous band of tiles: neighbouring tiles stage overlapping
    // source windows, which then hit that XCD's L2 instead of going out to the Infinity Cache.
    // (bijective for any tile count; affects speed only)
    const int ntile = gridDim.x;
    const int xcd = blockIdx.x & 7, qq = ntile >> 3, rr = ntile & 7;
    int tile = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (blockIdx.x >> 3);
    if (rr == 0 && qq % tiles_x == 0) {
        // the band is a whole number of tile rows: walk it column by column, so that the blocks in flight
        // on one XCD share a narrow strip of source columns (working set ~1 MB instead of the full image
        // width) -- measured HBM reads 485 MB -> see profiles/ (algorithmic 281 MB)
        const int band_rows = qq / tiles_x, i = blockIdx.x >> 3;
        tile = (xcd * band_rows + i % band_rows) * tiles_x + i / band_rows;
    }
    const int b = blockIdx.y;
    const int tx0 = (tile % tiles_x) * TW, ty0 = (tile / tiles_x) * TH;
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
    const auto ref_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)refb, (short)0, a.C * HW * 4, 0x00020000);
    const int nchunk = (a.C + 3) / 4;

    for (int v = 0; v < a.V; ++v) {
        ViewXform xf;
        make_view_xform(a.K + b * 9, a.R + ((size_t)b * a.V + v) * 9, a.t + ((size_t)b * a.V + v) * 3,
                        a.blas_mode, xf);
        float t2a, t2b, t2c;
        ray_term2(xf, r0, r1, r2, t2a, t2b, t2c);
        const float* srcv = a.src + (size_t)b * a.src_bstride + (size_t)v * a.src_vstride;
        const auto src_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)srcv, (short)0, a.C * HW * 4, 0x00020000);

        for (int k0 = 0; k0 < a.D; k0 += SG) {
            // ---- geometry of this thread's KP planes (registers) ----------------------------
            // Plane assignment inside the 32-plane super group: the 8 planes of a wave are split into
            // nsplit parts of per = 8/nsplit planes; part q of wave w holds planes
            //     k0 + q*(4*per) + w*per + [0, per),
            // so the planes staged together (one part of all 4 waves) are 4*per CONSECUTIVE planes and
            // their common window shrinks with nsplit (nsplit = 1 is simply planes 8w .. 8w+7).
            int off[KP];  // packed (y0, x0) until the plane's part is staged, then its window texel index
            int kpl[KP];  // depth plane of slot i
            float wnw[KP], wne[KP], wsw[KP], wse[KP];
            auto geometry = [&](int nsplit_) {
                const int per_ = KP / nsplit_;
#pragma unroll
                for (int i = 0; i < KP; ++i) {
                    kpl[i] = k0 + (i / per_) * (NPG * per_) + pgl * per_ + (i % per_);
                    const int k = min(kpl[i], a.D - 1);
                    float ix, iy;
                    plane_sample_pos(xf, t2a, t2b, t2c, a.d_candi[k], cx, cy, half_w, half_h, ix, iy);
                    const Footprint f = make_footprint(ix, iy, a.W, a.H);
                    wnw[i] = f.nw; wne[i] = f.ne; wsw[i] = f.sw; wse[i] = f.se;
                    if (f.mask != 0u && live) {  // at least one tap inside the image
                        off[i] = f.y0 * 65536 + (f.x0 & 0xffff);
                    } else {
                        // all four taps read zero: the sample is pointed at texel 0 of the window with
                        // weights that keep the reference's result (0 for finite positions, NaN for NaN
                        // positions because 0 * NaN = NaN as in ATen)
                        off[i] = INT_MIN;
                        wnw[i] = wnw[i] * 0.0f; wne[i] = wne[i] * 0.0f; wsw[i] = wsw[i] * 0.0f; wse[i] = wse[i] * 0.0f;
                    }
                    // two planes at a time: interleaving all 8 independent divide chains costs >100 VGPRs
                    if (i & 1) __builtin_amdgcn_sched_barrier(0);
                }
            };
            float acc[KP];
#pragma unroll
            for (int i = 0; i < KP; ++i) acc[i] = 0.0f;

            // ---- window of planes [first, first+count) of every wave: block bounding box ------------
            int wx0, wy0, wx1, wy1, WC, WR, ncb, nsub;
            int glog = 4;  // log2 of the staging grid width: the 256 threads walk the window as 16x16, 32x8 or 64x4
            bool empty;
            auto window_of = [&](int first, int count) -> bool {
                int bx0 = INT_MAX, by0 = INT_MAX, bx1 = INT_MIN, by1 = INT_MIN;
#pragma unroll
                for (int i = 0; i < KP; ++i) {
                    if (i >= first && i < first + count && off[i] != INT_MIN) {
                        const int fy0 = off[i] >> 16, fx0 = (int)(short)(off[i] & 0xffff);
                        bx0 = min(bx0, fx0); bx1 = max(bx1, fx0);
                        by0 = min(by0, fy0); by1 = max(by1, fy0);
                    }
                }
                bx0 = wave_min(bx0); by0 = wave_min(by0); bx1 = wave_max(bx1); by1 = wave_max(by1);
                __syncthreads();  // every reader of the previous s_bbox / window is done
                if (tid == 0) { s_bbox[0] = INT_MAX; s_bbox[1] = INT_MAX; s_bbox[2] = INT_MIN; s_bbox[3] = INT_MIN; }
                __syncthreads();
                if (lane == 0) {
                    atomicMin(&s_bbox[0], bx0); atomicMin(&s_bbox[1], by0);
                    atomicMax(&s_bbox[2], bx1); atomicMax(&s_bbox[3], by1);
                }
                __syncthreads();
                wx0 = s_bbox[0]; wy0 = s_bbox[1]; wx1 = s_bbox[2]; wy1 = s_bbox[3];
                empty = wx0 > wx1;  // every sample of these planes is fully out of bounds
                if (empty) { wx0 = 0; wx1 = 0; wy0 = 0; wy1 = 0; }
                WC = ((wx1 - wx0 + 2) + 15) & ~15;  // +1 east tap, pitch multiple of 16
                WR = wy1 - wy0 + 2;                  // +1 south tap
                // staging grid shape with the fewest sub-blocks (windows are usually wide and short)
                nsub = INT_MAX;
#pragma unroll
                for (int gl = 4; gl <= 6; ++gl) {
                    const int gw = 1 << gl, gh = 256 >> gl;
                    const int n = ((WR + gh - 1) / gh) * ((WC + gw - 1) >> gl);
                    if (n < nsub) { nsub = n; glog = gl; ncb = (WC + gw - 1) >> gl; }
                }
                return WC * WR <= NTEX_MAX && nsub <= SLOTS;  // block-uniform
            };
            // Smallest split into 1, 2 or 4 parts whose windows all fit LDS; the geometry is only redone
            // when the coarser split failed (large disparities per plane, e.g. 512x1024 with D=128).
            int nsplit = 1;
            for (;;) {
                geometry(nsplit);
                const int per = KP / nsplit;
                bool fits = true;
                for (int part = 0; part < nsplit && fits; ++part) fits = window_of(part * per, per);
                if (fits) break;
                if (nsplit == 4) {  // block-uniform: leave the tile to the gather kernel
                    if (tid == 0) tile_flags[b * ntile + tile] = 1;
                    return;
                }
                nsplit *= 2;
            }
            const int per = KP / nsplit;

            for (int part = 0; part < nsplit; ++part) {
            if (nsplit > 1) window_of(part * per, per);  // nsplit == 1: the fit check left this window
#pragma unroll
            for (int i = 0; i < KP; ++i) {
                if (i / per == part) {
                    const int fy0 = off[i] >> 16, fx0 = (int)(short)(off[i] & 0xffff);
                    off[i] = (off[i] == INT_MIN) ? 0 : (fy0 - wy0) * WC + (fx0 - wx0);
                }
            }

            // ---- channel chunks (4 channels each), software pipelined ------------------------
            // Staging uses raw buffer loads: a wave-uniform descriptor + scalar channel offset + one
            // per-lane texel offset that is loop invariant.  Out-of-image texels carry an offset
            // beyond the descriptor's range, for which the hardware returns 0 -- the zero padding.
            int so[SLOTS];
            {
                int rb = 0, cb = 0;
#pragma unroll
                for (int sl = 0; sl < SLOTS; ++sl) {
                    so[sl] = 0x7fffffff;
                    if (sl < nsub) {  // uniform
                        const int row = rb * (256 >> glog) + (tid >> glog), col = (cb << glog) + (tid & ((1 << glog) - 1));
                        const int gx = wx0 + col, gy = wy0 + row;
                        const bool inb = !empty && row < WR && col < WC && gx >= 0 && gx < a.W && gy >= 0 && gy < a.H;
                        so[sl] = inb ? (gy * a.W + gx) * 4 : 0x7fffffff;
                        if (++cb == ncb) { cb = 0; ++rb; }
                    }
                }
            }
            float4 st_w[SLOTS];
            float4 st_r;
            auto prefetch = [&](int ch) {
                const int c = ch * 4;
                const bool k1 = c + 1 < a.C, k2 = c + 2 < a.C, k3 = c + 3 < a.C;  // uniform channel tail
                const int s0 = c * HW * 4;
                const int s1 = k1 ? s0 + HW * 4 : s0, s2 = k2 ? s0 + 2 * HW * 4 : s0, s3 = k3 ? s0 + 3 * HW * 4 : s0;
#pragma unroll
                for (int sl = 0; sl < SLOTS; ++sl) {
                    if (sl < nsub) {  // uniform
                        const float v0 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(src_rsrc, so[sl], s0, 0));
                        const float v1 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(src_rsrc, so[sl], s1, 0));
                        const float v2 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(src_rsrc, so[sl], s2, 0));
                        const float v3 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(src_rsrc, so[sl], s3, 0));
                        st_w[sl] = make_float4(v0, v1, v2, v3);  // raw: no use of the data before commit()
                    }
                }
                if (pgl == 0) {  // wave 0 stages the tile's reference features
                    const float q0 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ref_rsrc, p * 4, s0, 0));
                    const float q1 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ref_rsrc, p * 4, s1, 0));
                    const float q2 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ref_rsrc, p * 4, s2, 0));
                    const float q3 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ref_rsrc, p * 4, s3, 0));
                    st_r = make_float4(q0, q1, q2, q3);
                }
            };
            auto commit = [&](int bufi, int ch) {
                // Channels beyond C were fetched from a clamped (valid) plane and must read as zero; that
                // only concerns the last chunk, and it is done here -- after the compute of the previous
                // chunk -- so that nothing waits on the loads while they fly.
                const int c = ch * 4;
                const bool tail = c + 3 >= a.C;  // uniform
                const bool k1 = c + 1 < a.C, k2 = c + 2 < a.C, k3 = c + 3 < a.C;
                float4* wb = win + bufi * NTEX_MAX;
                int rb = 0, cb = 0;
#pragma unroll
                for (int sl = 0; sl < SLOTS; ++sl) {
                    if (sl < nsub) {
                        const int row = rb * (256 >> glog) + (tid >> glog), col = (cb << glog) + (tid & ((1 << glog) - 1));
                        if (row < WR && col < WC) {
                            float4 val = st_w[sl];
                            if (tail) val = make_float4(val.x, k1 ? val.y : 0.f, k2 ? val.z : 0.f, k3 ? val.w : 0.f);
                            wb[row * WC + col] = val;
                        }
                        if (++cb == ncb) { cb = 0; ++rb; }
                    }
                }
                if (pgl == 0) {
                    float4 val = st_r;
                    if (tail) val = make_float4(val.x, k1 ? val.y : 0.f, k2 ? val.z : 0.f, k3 ? val.w : 0.f);
                    reft[bufi * 64 + lane] = val;
                }
            };
            prefetch(0);
            commit(0, 0);  // window_of() ended with a barrier: nobody reads the buffers any more
            __syncthreads();
            for (int ch = 0; ch < nchunk; ++ch) {
                const int cur = ch & 1;
                if (ch + 1 < nchunk) prefetch(ch + 1);  // in flight while this chunk is computed
                {
                    const float4 rf = reft[cur * 64 + lane];
                    const float4* wg = win + cur * NTEX_MAX;
#pragma unroll
                    for (int i = 0; i < KP; ++i) {
                        if (i / per != part) continue;  // uniform: this plane is staged in another part
                        // cap the taps in flight at two planes (32 VGPRs): left alone the scheduler hoists
                        // all 32 ds_read_b128 of the chunk and spills the geometry
                        if ((i & 1) == 0) __builtin_amdgcn_sched_barrier(0);
                        const float4 s00 = wg[off[i]];
                        const float4 s01 = wg[off[i] + 1];
                        const float4 s10 = wg[off[i] + WC];
                        const float4 s11 = wg[off[i] + WC + 1];
#define PDEPTH_TAP(comp)                                                            \
    {                                                                               \
        float diff = __builtin_fmaf(s00.comp, wnw[i], -rf.comp);                    \
        diff = __builtin_fmaf(s01.comp, wne[i], diff);                              \
        diff = __builtin_fmaf(s10.comp, wsw[i], diff);                              \
        diff = __builtin_fmaf(s11.comp, wse[i], diff);                              \
        acc[i] = METRIC == 0 ? __builtin_fmaf(diff, diff, acc[i]) : acc[i] + fabsf(diff); \
    }
                        PDEPTH_TAP(x) PDEPTH_TAP(y) PDEPTH_TAP(z) PDEPTH_TAP(w)
#undef PDEPTH_TAP
                        // pin the accumulation here: IR-level sinking otherwise moves the fma chains of all
                        // 8 planes behind the last load and keeps 128 tap registers alive
                        if (i & 1) asm volatile("" : "+v"(acc[i - 1]), "+v"(acc[i]));
                    }
                }
                if (ch + 1 < nchunk) commit(cur ^ 1, ch + 1);
                __syncthreads();
            }
            }  // parts
            // channels beyond C were staged as zeros on both sides: they add (0-0)^2 = 0, except for
            // NaN-weight samples where they add NaN -- which the reference produces as well.
#pragma unroll
            for (int i = 0; i < KP; ++i) {
                if (kpl[i] < a.D) {
                    float* o = costs + (size_t)kpl[i] * 64 + lane;  // owned by this thread only
                    const float c = acc[i] / a.sigma;
                    *o = (v == 0) ? (0.0f + c) : (*o + c);
                }
            }
        }
    }
    __syncthreads();

    // ---- epilogue from LDS: cost store, log-softmax over D, expectation ----------------------
    // wave w handles planes k = w, w+4, w+8, ... of the tile's 64 pixels
    float* cout = (a.cost_out && live) ? a.cost_out + (size_t)b * a.D * HW + p : nullptr;
    if (cout)
        for (int k = pgl; k < a.D; k += NPG) cout[(size_t)k * HW] = costs[k * 64 + lane];
    if (a.logp_out || a.depth_out) {
        float m = -INFINITY;
        for (int k = pgl; k < a.D; k += NPG) m = fmaxf(m, costs[k * 64 + lane]);
        red[pgl * 64 + lane] = m;
        __syncthreads();
        m = fmaxf(fmaxf(red[lane], red[64 + lane]), fmaxf(red[128 + lane], red[192 + lane]));
        __syncthreads();
        float s = 0.0f;
        for (int k = pgl; k < a.D; k += NPG) s = s + expf(costs[k * 64 + lane] - m);
        red[pgl * 64 + lane] = s;
        __syncthreads();
        s = (red[lane] + red[64 + lane]) + (red[128 + lane] + red[192 + lane]);
        __syncthreads();
        const float ls = logf(s);
        float e = 0.0f;
        float* o = (a.logp_out && live) ? a.logp_out + (size_t)b * a.D * HW + p : nullptr;
        for (int k = pgl; k < a.D; k += NPG) {
            const float lp = (costs[k * 64 + lane] - m) - ls;
            if (o) o[(size_t)k * HW] = lp;
            e = e + a.d_candi[k] * expf(lp);
        }
        if (a.depth_out) {
            red[pgl * 64 + lane] = e;
            __syncthreads();
            if (pgl == 0 && live)
                a.depth_out[(size_t)b * HW + p] = (red[lane] + red[64 + lane]) + (red[128 + lane] + red[192 + lane]);
        }
    }
}

static size_t tiled_lds_bytes(int D) {
    return (size_t)(NBUF * NTEX_MAX + NBUF * 64) * sizeof(float4) + (size_t)(D + NPG) * 64 * sizeof(float);
}

// Largest D whose cost tile fits LDS next to the window (2 blocks per CU).
int sweep_tiled_max_planes() { return 160; }

size_t sweep_tiled_workspace_bytes(int B, int H, int W) {
    const size_t tiles = (size_t)((W + TW - 1) / TW) * ((H + TH - 1) / TH);
    return ((size_t)B * tiles * sizeof(int) + 255) & ~(size_t)255;
}

// Launches the tiled kernel, then the gather kernel on the tiles it flagged.
hipError_t launch_sweep_tiled(const SweepArgs& a, void* workspace, hipStream_t stream) {
    const int tiles_x = (a.W + TW - 1) / TW, tiles_y = (a.H + TH - 1) / TH;
    const int tiles = tiles_x * tiles_y;
    int* flags = reinterpret_cast<int*>(workspace);
    hipError_t e = hipMemsetAsync(flags, 0, (size_t)a.B * tiles * sizeof(int), stream);
    if (e != hipSuccess) return e;
    const size_t lds = tiled_lds_bytes(a.D);
    dim3 grid(tiles, a.B);
    if (a.metric == 0) {
        auto kern = sweep_tiled_kernel<0>;
        if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(kern, grid, dim3(256), lds, stream, a, flags, tiles_x);
    } else {
        auto kern = sweep_tiled_kernel<1>;
        if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(kern, grid, dim3(256), lds, stream, a, flags, tiles_x);
    }
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    return launch_sweep_direct_flagged(a, flags, tiles_x, tiles, stream);
}

}  // namespace pdepth
