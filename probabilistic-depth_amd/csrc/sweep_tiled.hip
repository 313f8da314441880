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
//   source  = a pre-pass (pack_c4_kernel, HBM-rate) re-lays every source view from NCHW to channel-group-
//             planar [C/4][H][W] float4 texels in the workspace, zero padded to a multiple of 4 channels;
//   window  = bounding box of every tap the block touches in the current super group, staged four
//             channels at a time into LDS as float4 texels [row][col] by LDS-DMA (buffer_load_dwordx4 ...
//             lds: a wave-instruction moves 64 texels = 1 KiB, no registers, no ds_write), double
//             buffered: chunk ch+1 is in flight while chunk ch is computed; one barrier per chunk.  The
//             DMA is issued from inline asm with hand-counted s_waitcnt, because the compiler would
//             otherwise drain it in front of every ds_read.  Texels outside the image are fetched with an
//             out-of-range buffer offset and arrive as zeros, which IS padding_mode='zeros' -- no per-tap
//             masks in the inner loop.  The row pitch is a multiple of 16 texels;
//   taps    = software pipelined over the wave's planes: the four ds_read_b128 of plane i+1 are in flight
//             while the 20 fma of plane i execute; the window buffer is an immediate of the ds_read;
//   ref     = the tile's reference features of the chunk (wave w moves channel 4*ch+w by LDS-DMA from NCHW);
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
constexpr int NBUF = 2;           // LDS window buffers: chunk ch+1 is in flight (LDS-DMA) while chunk ch is computed
#ifndef PDEPTH_NTEX   // experiment knobs (tools/variants.sh): window texels per buffer, blocks per CU
#define PDEPTH_NTEX 1024
#endif
#ifndef PDEPTH_OCC
#define PDEPTH_OCC 3
#endif
constexpr int NTEX_MAX = PDEPTH_NTEX;    // window texels per buffer (LDS: NTEX_MAX*NBUF*16 B = 32 KB)
constexpr int SLOTS = (PDEPTH_NTEX + 255) / 256;          // sub-blocks of a window (256 texels each = one DMA pass of the block)

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

typedef int v4i __attribute__((ext_vector_type(4)));
typedef float v4f __attribute__((ext_vector_type(4)));

// LDS-DMA issued from inline asm: the compiler must not know that these loads write LDS, or it drains them
// (s_waitcnt vmcnt(0)) in front of the next ds_read of the other buffer and nothing overlaps.  The waits are
// counted by hand (wait_vm) in front of the barrier that publishes a buffer.
__device__ __forceinline__ void dma_b128(v4i rsrc, unsigned lds_addr, int voff, int soff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                 :: "s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff) : "memory", "m0");
}
__device__ __forceinline__ void dma_b32(v4i rsrc, unsigned lds_addr, int voff, int soff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dword %1, %2, %3 offen lds"
                 :: "s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff) : "memory", "m0");
}
__device__ __forceinline__ void wait_dma() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// raw barrier: every LDS access of this wave has completed, but VMEM (the DMA of later chunks) stays in flight
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
__device__ __forceinline__ v4i make_rsrc(const void* base, int bytes) {
    const unsigned long long p = reinterpret_cast<unsigned long long>(base);
    v4i r;
    r.x = (int)(unsigned)p; r.y = (int)(unsigned)(p >> 32) & 0xffff; r.z = bytes; r.w = 0x00020000;
    return r;
}
__device__ __forceinline__ unsigned lds_addr_of(const void* p) {
    return (unsigned)(unsigned long long)(__attribute__((address_space(3))) const void*)p;
}

}  // namespace

template <int METRIC>
__global__ __launch_bounds__(256, PDEPTH_OCC) void sweep_tiled_kernel(SweepArgs a, const float4* __restrict__ packed,
                                                              int* __restrict__ tile_flags, int tiles_x) {
    extern __shared__ __attribute__((aligned(16))) float4 lds4[];
    float4* win = lds4;                                   // [NBUF][NTEX_MAX]
    float* reft = reinterpret_cast<float*>(lds4 + NBUF * NTEX_MAX);  // [NBUF][4 channels][64 pixels]
    float* costs = reft + NBUF * 256;                     // [D][64]
    float* red = costs + (size_t)a.D * 64;                // [NPG][64]
    float* dcl = red + NPG * 64;                          // [D] depth candidates (read wave-uniformly, per plane)
    __shared__ int s_bbox[4];

    const int tid = threadIdx.x;
    const int pgl = __builtin_amdgcn_readfirstlane(tid >> 6);  // plane group of this wave (wave-uniform)
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
    const float rcx = refined_rcp(cx), rcy = refined_rcp(cy);
    const float r0 = a.rays[((size_t)b * 3 + 0) * HW + p];
    const float r1 = a.rays[((size_t)b * 3 + 1) * HW + p];
    const float r2 = a.rays[((size_t)b * 3 + 2) * HW + p];
    const float* refb = a.ref + (size_t)b * a.ref_bstride;
    const v4i ref_rsrc = make_rsrc(refb, a.C * HW * 4);
    const int nchunk = (a.C + 3) / 4;
    for (int k = tid; k < a.D; k += 256) dcl[k] = a.d_candi[k];
    __syncthreads();
    const int win_lds0 = (int)lds_addr_of(win);

    for (int v = 0; v < a.V; ++v) {
        ViewXform xf;
        make_view_xform(a.K + b * 9, a.R + ((size_t)b * a.V + v) * 9, a.t + ((size_t)b * a.V + v) * 3,
                        a.blas_mode, xf);
        float t2a, t2b, t2c;
        ray_term2(xf, r0, r1, r2, t2a, t2b, t2c);
        const float4* srcv = packed + ((size_t)b * a.V + v) * nchunk * HW;  // [C/4][H][W] float4 texels
        const v4i src_rsrc = make_rsrc(srcv, nchunk * HW * 16);

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
                const int lper = nsplit_ == 1 ? 3 : nsplit_ == 2 ? 2 : 1;  // log2(planes per part and wave)
#pragma unroll
                for (int i = 0; i < KP; ++i) {
                    kpl[i] = k0 + ((i >> lper) << (lper + 2)) + (pgl << lper) + (i & ((1 << lper) - 1));
                    const int k = min(kpl[i], a.D - 1);
                    float ix, iy;
                    plane_sample_pos_fast(xf, t2a, t2b, t2c, dcl[k], cx, cy, rcx, rcy, half_w, half_h, ix, iy);
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
            int wx0, wy0, wx1, wy1, WC, WR;
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
                return WC * WR <= NTEX_MAX;   // block-uniform
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
                    // LDS byte address of the top-left tap in window buffer 0
                    off[i] = win_lds0 + ((off[i] == INT_MIN) ? 0 : (fy0 - wy0) * WC + (fx0 - wx0)) * 16;
                }
            }

            // ---- channel chunks (4 channels each), double buffered through LDS-DMA ------------
            // The window is staged by buffer_load_dwordx4 ... lds: a wave-instruction moves 64 texels
            // (1 KiB) from the packed source straight into 64 consecutive float4 slots of the LDS
            // window, no registers involved.  Per lane only the texel's byte offset is needed, and it is
            // invariant over the chunks (the channel group is the scalar offset).  Out-of-image texels
            // carry an offset beyond the descriptor's range, for which the hardware delivers 0 -- the
            // zero padding.
            int so[SLOTS];
#pragma unroll
            for (int sl = 0; sl < SLOTS; ++sl) {
                const int tl = sl * 256 + tid;
                const int row = tl / WC, col = tl - row * WC;
                const int gx = wx0 + col, gy = wy0 + row;
                const bool inb = !empty && row < WR && gx >= 0 && gx < a.W && gy >= 0 && gy < a.H;
                so[sl] = inb ? (gy * a.W + gx) * 16 : 0x7fffffff;
            }
            // reference features: wave w moves channel 4*ch + w of the tile's 64 pixels
            const int ro = p * 4;
            const unsigned win_lds = lds_addr_of(win + pgl * 64), ref_lds = lds_addr_of(reft + pgl * 64);
            auto stage = [&](int bufi, int ch) {
                const int soff = ch * HW * 16;
#pragma unroll
                for (int sl = 0; sl < SLOTS; ++sl)
                    if (sl * 256 + pgl * 64 < WC * WR)  // wave-uniform: this wave's 64 texels are part of the window
                        dma_b128(src_rsrc, win_lds + (bufi * NTEX_MAX + sl * 256) * 16, so[sl], soff);
                const int c = ch * 4 + pgl;
                // channels beyond C: out of range => zeros (the packed source is zero padded as well)
                dma_b32(ref_rsrc, ref_lds + bufi * 1024, c < a.C ? ro : 0x7fffffff, c < a.C ? c * HW * 4 : 0);
            };
            // Iteration ch: wait until this wave's DMA of chunk ch has landed, barrier (=> every wave's part of
            // chunk ch is in LDS and every wave is done reading chunk ch-1), re-fill the buffer of chunk ch-1
            // with chunk ch+1, compute chunk ch.  The two buffers get their own copy of the compute code so
            // that the buffer base is an immediate of the ds_read (no per-chunk address arithmetic).
            typedef const __attribute__((address_space(3))) v4f* lds_v4f;
            const int WCB = WC * 16;
            lds_barrier();  // every reader of the previous part / super group is done with the buffers
            stage(0, 0);
            for (int ch = 0; ch < nchunk; ++ch) {
                wait_dma();
                lds_barrier();
                if (ch + 1 < nchunk) stage((ch + 1) & 1, ch + 1);
                {
#define PDEPTH_TAP(T, comp)                                                         \
    {                                                                               \
        float diff = __builtin_fmaf(T[0].comp, wnw[i], -rf.comp);                   \
        diff = __builtin_fmaf(T[1].comp, wne[i], diff);                             \
        diff = __builtin_fmaf(T[2].comp, wsw[i], diff);                             \
        diff = __builtin_fmaf(T[3].comp, wse[i], diff);                             \
        acc[i] = METRIC == 0 ? __builtin_fmaf(diff, diff, acc[i]) : acc[i] + fabsf(diff); \
    }
#define PDEPTH_LOAD(T, i_)                                                          \
    {                                                                               \
        const lds_v4f q0 = (lds_v4f)(size_t)(unsigned)(off[i_] + CUR * NTEX_MAX * 16);        \
        const lds_v4f q1 = (lds_v4f)(size_t)(unsigned)(off[i_] + WCB + CUR * NTEX_MAX * 16);  \
        T[0] = q0[0]; T[1] = q0[1]; T[2] = q1[0]; T[3] = q1[1];                     \
    }
                    // Software pipeline over the part's planes: the four taps of plane i+1 are in flight while
                    // the 20 fma of plane i execute (two tap sets = 32 VGPRs).  The sched_barriers keep the
                    // compiler from hoisting more loads (spills) or sinking the fma chains.
#define PDEPTH_COMPUTE(PER, BASE)                                                                   \
    {                                                                                               \
        v4f ta[4], tb[4];                                                                           \
        PDEPTH_LOAD(ta, BASE)                                                                       \
        _Pragma("unroll") for (int i = BASE; i < BASE + PER; ++i) {                                 \
            if (i + 1 < BASE + PER) {                                                               \
                if ((i - BASE) & 1) PDEPTH_LOAD(ta, i + 1) else PDEPTH_LOAD(tb, i + 1)              \
            }                                                                                       \
            __builtin_amdgcn_sched_barrier(0);                                                      \
            if ((i - BASE) & 1) { PDEPTH_TAP(tb, x) PDEPTH_TAP(tb, y) PDEPTH_TAP(tb, z) PDEPTH_TAP(tb, w) } \
            else { PDEPTH_TAP(ta, x) PDEPTH_TAP(ta, y) PDEPTH_TAP(ta, z) PDEPTH_TAP(ta, w) }        \
            asm volatile("" : "+v"(acc[i]));                                                        \
            __builtin_amdgcn_sched_barrier(0);                                                      \
        }                                                                                           \
    }
                    // (plane slots of a part and the buffer are compile-time constants in every instantiation)
#define PDEPTH_CHUNK(CUR)                                                                           \
    {                                                                                               \
        const float* rp = reft + CUR * 256 + lane;                                                  \
        const float4 rf = make_float4(rp[0], rp[64], rp[128], rp[192]);                             \
        if (per == 8) PDEPTH_COMPUTE(8, 0)                                                          \
        else if (per == 4) { if (part == 0) PDEPTH_COMPUTE(4, 0) else PDEPTH_COMPUTE(4, 4) }        \
        else if (part == 0) PDEPTH_COMPUTE(2, 0)                                                    \
        else if (part == 1) PDEPTH_COMPUTE(2, 2)                                                    \
        else if (part == 2) PDEPTH_COMPUTE(2, 4)                                                    \
        else PDEPTH_COMPUTE(2, 6)                                                                   \
    }
                    if (ch & 1) {
                        constexpr int CUR = 1;
                        PDEPTH_CHUNK(CUR)
                    } else {
                        constexpr int CUR = 0;
                        PDEPTH_CHUNK(CUR)
                    }
#undef PDEPTH_CHUNK
#undef PDEPTH_COMPUTE
#undef PDEPTH_TAP
#undef PDEPTH_LOAD
                }
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
            e = e + dcl[k] * expf(lp);
        }
        if (a.depth_out) {
            red[pgl * 64 + lane] = e;
            __syncthreads();
            if (pgl == 0 && live)
                a.depth_out[(size_t)b * HW + p] = (red[lane] + red[64 + lane]) + (red[128 + lane] + red[192 + lane]);
        }
    }
}

// NCHW -> channel-group-planar [C/4][H][W] float4 (channels beyond C are zero): what the sweep kernel stages
// with 16-byte LDS-DMA.  One thread per (pixel, channel group); reads are 256-byte and writes 1-KiB wave
// transactions.
__global__ __launch_bounds__(256) void pack_c4_kernel(const float* __restrict__ src, long long bstride,
                                                      long long vstride, int V, int C, int HW,
                                                      float4* __restrict__ out) {
    const int pix = blockIdx.x * 256 + threadIdx.x;
    if (pix >= HW) return;
    const int g = blockIdx.y, bv = blockIdx.z;
    const float* s = src + (size_t)(bv / V) * bstride + (size_t)(bv % V) * vstride + (size_t)g * 4 * HW + pix;
    const int c = g * 4;
    float4 o;
    o.x = s[0];
    o.y = c + 1 < C ? s[HW] : 0.0f;
    o.z = c + 2 < C ? s[2 * (size_t)HW] : 0.0f;
    o.w = c + 3 < C ? s[3 * (size_t)HW] : 0.0f;
    out[((size_t)bv * gridDim.y + g) * HW + pix] = o;
}

static size_t tiled_lds_bytes(int D) {
    return (size_t)(NBUF * NTEX_MAX + NBUF * 64) * sizeof(float4) + (size_t)(D + NPG) * 64 * sizeof(float) +
           (size_t)D * sizeof(float);
}

// Largest D whose cost tile fits LDS next to the window (2 blocks per CU).

// Largest D whose cost tile fits LDS next to the window buffers (2 blocks per CU).
int sweep_tiled_max_planes() { return 160; }

static size_t flag_bytes(int B, int H, int W) {
    const size_t tiles = (size_t)((W + TW - 1) / TW) * ((H + TH - 1) / TH);
    return ((size_t)B * tiles * sizeof(int) + 255) & ~(size_t)255;
}
size_t sweep_tiled_workspace_bytes(int B, int V, int C, int H, int W) {
    return flag_bytes(B, H, W) + (size_t)B * V * ((C + 3) / 4) * H * W * sizeof(float4);
}

// Launches the tiled kernel, then the gather kernel on the tiles it flagged.
hipError_t launch_sweep_tiled(const SweepArgs& a, void* workspace, hipStream_t stream) {
    const int tiles_x = (a.W + TW - 1) / TW, tiles_y = (a.H + TH - 1) / TH;
    const int tiles = tiles_x * tiles_y;
    int* flags = reinterpret_cast<int*>(workspace);
    float4* packed = reinterpret_cast<float4*>(static_cast<char*>(workspace) + flag_bytes(a.B, a.H, a.W));
    {
        const int HW = a.H * a.W;
        dim3 pgrid((HW + 255) / 256, (a.C + 3) / 4, a.B * a.V);
        hipLaunchKernelGGL(pack_c4_kernel, pgrid, dim3(256), 0, stream, a.src, a.src_bstride, a.src_vstride, a.V, a.C, HW, packed);
    }
    hipError_t e = hipMemsetAsync(flags, 0, (size_t)a.B * tiles * sizeof(int), stream);
    if (e != hipSuccess) return e;
    const size_t lds = tiled_lds_bytes(a.D);
    dim3 grid(tiles, a.B);
    if (a.metric == 0) {
        auto kern = sweep_tiled_kernel<0>;
        if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(kern, grid, dim3(256), lds, stream, a, packed, flags, tiles_x);
    } else {
        auto kern = sweep_tiled_kernel<1>;
        if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(kern, grid, dim3(256), lds, stream, a, packed, flags, tiles_x);
    }
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    return launch_sweep_direct_flagged(a, flags, tiles_x, tiles, stream);
}

}  // namespace pdepth
