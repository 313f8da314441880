// Cell-list plane sweep: the fast path of pdepth_sweep_{cost,dpv}_f32 for the L2 metric (round 2).
//
// Replaces est_swp_volume_v4 / _back_warp_homo_parallel / img_dis_L2_pard (warping/homography.py:98-135,
// :170-198, :80-82) and the log_softmax + dpv_to_depthmap tail (models/packnet.py:394, utils/img_utils.py:52-61).
//
// Idea.  Along the depth sweep the sample of one reference pixel walks down its epipolar line, and many
// consecutive planes fall into the SAME bilinear cell (2x2 source texels): on the benchmark poses a pixel touches
// 18 (mono) / 22 (stereo) distinct cells with its 64 planes.  With the bilinear weights w_t of plane k in cell i
//     sum_c (sum_t w_t s_t[c] - r[c])^2 = w^T G_i w - 2 sum_t w_t X_{i,t} + |r|^2,      X_{i,t} = <r, s_t>,
// so the channel loop is needed ONCE PER CELL (4 dot products) instead of once per plane, and the Gram terms G
// of neighbouring source texels depend on the source only (pre-pass).  Unlike the bounding-box band mode of
// sweep_tiled.hip there is no geometric assumption: the cell list of a pixel is exact, a pixel whose every plane
// has its own cell simply degenerates to 4 dot products per plane (cheaper than the 5 ops/channel of direct
// evaluation), and no plane is ever evaluated directly.
//
//   lanes   = 4 lanes per reference pixel (a quad), 16 pixels per wave (one row of the 16x4 tile), 4 waves per
//             block.  In the channel loop lane g of a quad owns corner g of every cell (X accumulators in
//             registers, one ds_read_b128 + 4 fma per cell and 4 channels); in the plane loop it owns planes
//             k = 4j + g (costs in registers, statically indexed).  Everything a pixel needs from its three
//             sibling lanes moves by DPP quad permutes; nothing is exchanged between waves.
//   X dump  = after the channel loop a wave writes its X to a wave-private LDS array [slot][pixel][corner]
//             (aliasing the window ring, which is dead by then) and every plane reads the four X of its cell
//             with one ds_read_b128 at a per-lane slot -- LDS is the only storage a lane can index dynamically.
//   window  = bounding box of the cells of the block's 64 pixels, staged 4 channels at a time by LDS-DMA from the
//             channel-group-planar copy of the source (pack_c4_kernel), double buffered, one barrier per chunk;
//             the two Gram planes of the window are staged once per pass next to the ring.
//   ref     = one buffer_load_dword per lane and chunk (lane g loads channel 4 ch + g of its pixel, the quad
//             shares the four values by DPP), prefetched one chunk ahead.
//   passes  = the planes are handled in windows of 64 (16 per lane); a window is cut into passes of consecutive
//             plane steps whenever a pixel would need more than NS = 32 cell slots or the staged window would
//             exceed WT texels.  The benchmark configurations run one pass per view; D = 128 at 512x1024 two to
//             four.  A tile whose single plane step does not fit is flagged for the gather kernel.
//   grid    = persistent blocks pulling (batch item, tile) work items from per-XCD queues, a contiguous band of
//             tiles per XCD walked column by column (as sweep_tiled.hip).  The mapping assumes the SPX partition
//             mode (workgroups dealt round-robin over the 8 XCDs); any other placement only costs L2 locality.
//
// LDS per block: 32 KB ring / dump + 20 KB Gram window + depth candidates = 53 KB -> 3 blocks (12 waves) per CU.
// Sample positions, weights and zero padding are those of geometry.hpp (bit-faithful to the reference's CPU
// path); the correlation form agrees with direct evaluation to an ulp or two of the cost (tests/test_hip_parity.py).
#include <hip/hip_runtime.h>

#include "cells_common.hpp"
#include "../geometry.hpp"
#include "../kernels.hpp"

namespace pdepth {

namespace {

constexpr int TW = 16, TH = 4;  // tile of reference pixels per block; wave w owns row w
constexpr int NW = 4;           // waves per block
constexpr int NT = 64 * NW;
constexpr int NS = 32;          // cell slots per pixel and pass
constexpr int JS = 16;          // plane steps per lane and window: a window is 4 * JS = 64 planes
// Build knobs (tools/variants_cells.sh): blocks per CU the register allocation must allow, window texels per ring
// buffer, ring depth.  Default: 3 blocks per CU, 53.6 KB of LDS per block.
#ifndef CELLS_OCC
#define CELLS_OCC 2
#endif
#ifndef CELLS_WT
#define CELLS_WT (CELLS_OCC >= 3 ? 704 : 1024)
#endif
#ifndef CELLS_KEEP_POS
#define CELLS_KEEP_POS 0
#endif
#ifndef CELLS_SETS   // register sets of 4 cells in the channel loop: SETS - 1 groups of reads in flight
#define CELLS_SETS 2
#endif
#ifndef CELLS_NBUF
#define CELLS_NBUF 3
#endif
constexpr int WT = CELLS_WT;     // window texels per ring buffer (multiple of 64)
constexpr int SETS = CELLS_SETS;
constexpr int NBUF = CELLS_NBUF; // window ring: chunks ch+1 .. ch+NBUF-1 are in flight while chunk ch is computed
constexpr int SLOTS = (WT + NT - 1) / NT;  // DMA instructions per wave, chunk and buffer (64 texels each)
constexpr int BUF_BYTES = WT * 16;
constexpr int RING_BYTES = NBUF * BUF_BYTES;     // window buffers; aliased by the X dumps [NW][NS][16 px][4]
constexpr int DUMP_WAVE_BYTES = NS * 256;
constexpr int CLIST_WAVE_BYTES = 16 * NS * 4;    // cell lists [16 px][NS] of a wave
constexpr int GRAMA_OFF = RING_BYTES;            // [WT] float4 (N, H, V, D1)
constexpr int GRAMB_OFF = GRAMA_OFF + WT * 16;   // [WT] float   D2
constexpr int REF_OFF = GRAMB_OFF + WT * 4;      // [NBUF][NW][64] reference features of the chunks in flight
constexpr int DTAB_OFF = REF_OFF + NBUF * NW * 256;  // [D] depth candidates
static_assert(NS == 32, "the channel loop is written out for 8 groups of 4 cells");
static_assert(WT % 64 == 0 && SLOTS <= 4, "whole DMA instructions, at most 5 per stage (wait_but)");
static_assert(NW * DUMP_WAVE_BYTES <= RING_BYTES, "X dumps must fit the window ring");
static_assert(NW * CLIST_WAVE_BYTES <= BUF_BYTES, "cell lists must fit the last ring buffer");
static_assert(RING_BYTES + 1024 <= 65536, "packed 16-bit LDS addresses of the window taps");
static_assert(CELLS_OCC * (DTAB_OFF + 512 + 320) <= 160 * 1024, "LDS of CELLS_OCC blocks per CU");
static_assert(NBUF == 2 || NBUF == 3, "ring depth");

constexpr int KEY_NONE = INT_MIN;  // cell key of a plane without any tap inside the image

}  // namespace

// NWIN = number of 64-plane windows (1: D <= 64, 2: D <= 128); the costs of a lane live in NWIN * 16 registers.
// MV = more than one source view: the costs then stay live across the channel loops of the later views (V = 1 has
// its own instantiation so that they never are).
// FAST = the straight-line instantiation: D = 64 NWIN exactly and one pass per window and view; a tile that needs
// more (cell slots, window texels) is flagged 1 and redone by the generic instantiation, which then runs on the list of
// flagged tiles only (redo_list, count in queue[0] of its own counters) and flags what it cannot do either with 2, for
// the gather kernel.  Straight-line matters: with
// the uniform guards of the generic code between the steps, hipcc drains lgkmcnt at every join and nothing overlaps.
template <int NWIN, bool MV, bool FAST>
__global__ __launch_bounds__(NT, CELLS_OCC) void sweep_cells_kernel(SweepArgs a, const float4* __restrict__ packed,
                                                            int* __restrict__ tile_flags, int* __restrict__ queue,
                                                            int tiles_x, int ntile, const int* __restrict__ redo_list) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    __shared__ int s_ex[2][NW][8];  // per-wave values of a block-wide reduction, double buffered by parity
    __shared__ int s_item[2];       // work item of this block: current / next
    int ex_parity = 0;

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int q = lane >> 2;   // pixel of the wave's row
    const int g = lane & 3;    // role in the quad: corner (channel loop), plane residue (plane loop)
    const unsigned lds0 = lds_addr_of(lds_raw);
    float* dtab = reinterpret_cast<float*>(lds_raw + DTAB_OFF);
    const unsigned dump0 = lds0 + wave * DUMP_WAVE_BYTES;  // this wave's X dump (inside the ring)
    // this wave's cell lists [16 pixels][NS]: in the last ring buffer, which is dead between the block-wide exchange
    // of a pass (every wave is past its previous plane loop) and the barrier of chunk 0 (after which chunk NBUF-1 is
    // staged into it)
    const unsigned clist0 = lds0 + (NBUF - 1) * BUF_BYTES + wave * CLIST_WAVE_BYTES;

    const int HW = a.H * a.W;
    const int nchunk = (a.C + 3) / 4;
    const float half_w = (float)a.W / 2.0f, half_h = (float)a.H / 2.0f;
    const float sigma = a.sigma, rsigma = refined_rcp(sigma);
    auto div_sigma = [&](float v) {
        float r = div_core(v, sigma, rsigma);
        if (!(fabsf(v) < 1.0e30f)) r = v / sigma;  // inf / NaN exactly like the IEEE divide
        return r;
    };
    for (int k = tid; k < a.D; k += NT) dtab[k] = CELLS_ARG(const float*, d_candi)[k];
    __syncthreads();

    // ---- persistent work loop (XCD-aware item order, see sweep_tiled.hip) ---------------------------------
    const int xcd = blockIdx.x & 7, qq = ntile >> 3, rr8 = ntile & 7;
    const int band_first = xcd < rr8 ? xcd * (qq + 1) : rr8 * (qq + 1) + (xcd - rr8) * qq;
    const int band_tiles = qq + (xcd < rr8 ? 1 : 0);
    const int nitems = band_tiles * a.B;
    const bool colmajor = rr8 == 0 && qq % tiles_x == 0;
    const bool listed = redo_list != nullptr;   // work items = entries of the list the fast kernel wrote
    const bool queued = !listed && (int)gridDim.x < 8 * ((ntile + 7) / 8) * a.B;
    if (queued) {
        if (tid == 0) s_item[0] = atomicAdd(&queue[xcd], 1);
        __syncthreads();
    }
    const int nwork = listed ? *(const volatile int*)&queue[0] : nitems;
    int item = listed ? (int)blockIdx.x : queued ? s_item[0] : (int)(blockIdx.x >> 3), item_par = 0;
#ifdef CELLS_STAMPS
    unsigned long long stamp_acc[13] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long stamp_t = __builtin_amdgcn_s_memtime();
#endif
    while (item < nwork) {
        CELLS_STAMP(9)
        if (tid == 0) s_item[item_par ^ 1] = listed ? item + (int)gridDim.x : queued ? atomicAdd(&queue[xcd], 1) : nwork;
        int b, tile;
        if (listed) {
            const int id = redo_list[item];
            b = id / ntile;
            tile = id - b * ntile;
        } else {
            b = item / band_tiles;
            const int ti = item - b * band_tiles;
            tile = band_first + ti;
            if (colmajor) {
                const int band_rows = qq / tiles_x;
                tile = (xcd * band_rows + ti % band_rows) * tiles_x + ti / band_rows;
            }
        }
        int* const my_flag = tile_flags + (b * ntile + tile);
        const int px = (tile % tiles_x) * TW + q, py = (tile / tiles_x) * TH + wave;
        const bool live = px < a.W && py < a.H;
        const int p = min(py, a.H - 1) * a.W + min(px, a.W - 1);

        const float* cxcy_ = CELLS_ARG(const float*, cxcy);
        const float cx = cxcy_[b * 2 + 0], cy = cxcy_[b * 2 + 1];
        const float rcx = refined_rcp(cx), rcy = refined_rcp(cy);
        const float* rays_ = CELLS_ARG(const float*, rays) + (size_t)b * 3 * HW + p;
        const float r0 = rays_[0], r1 = rays_[HW], r2 = rays_[2 * (size_t)HW];

        CELLS_STAMP(0)  // tile start: work item, camera, rays
        // cost of plane 64 m + 4 j + g at [m * JS + j]; first written by the plane loops of view 0 (every plane belongs
        // to exactly one pass), so for V = 1 no cost register is live across a channel loop
        float cost[NWIN * JS];
        bool bail = false;  // block-uniform: the tile goes to the gather kernel

        for (int v = 0; v < (MV ? a.V : 1) && !bail; ++v) {
            ViewXform xf;
            make_view_xform(CELLS_ARG(const float*, K) + b * 9, CELLS_ARG(const float*, R) + ((size_t)b * a.V + v) * 9,
                            CELLS_ARG(const float*, t) + ((size_t)b * a.V + v) * 3, CELLS_ARG(int, blas_mode), xf);
            float t2a, t2b, t2c;
            ray_term2(xf, r0, r1, r2, t2a, t2b, t2c);
            const float4* srcv = packed + ((size_t)b * a.V + v) * (nchunk + 2) * HW;
            const v4i src_rsrc = make_rsrc(srcv, (nchunk + 2) * HW * 16);
            const v4i ref_rsrc = make_rsrc(CELLS_ARG(const float*, ref) + (size_t)b * CELLS_ARG(long long, ref_bstride), a.C * HW * 4);

#pragma unroll
            for (int m = 0; m < NWIN; ++m) {
                const int kw = m * 4 * JS;       // first plane of the window
                if (kw >= a.D || bail) break;    // uniform
                const int nstep = FAST ? JS : min(JS, (a.D - kw + 3) >> 2);  // plane steps of this window

                // ---- positions of this lane's planes of the window -----------------------------------
#if CELLS_KEEP_POS
                float fx[JS], fy[JS];  // fractional sample position inside the cell
#endif
                int ki[JS];            // cell key (y0 << 16 | x0 & 0xffff) or KEY_NONE; later the packed plane info
                const int gk = opaque_v(kw + g);
#pragma unroll
                for (int j = 0; j < JS; ++j) {
                    const int k = gk + 4 * j;
                    float ix, iy;
                    plane_sample_pos_fast(xf, t2a, t2b, t2c, dtab[min(k, a.D - 1)], cx, cy, rcx, rcy, half_w, half_h, ix, iy);
                    const float xfl = floorf(ix), yfl = floorf(iy);
#if CELLS_KEEP_POS
                    fx[j] = ix - xfl;
                    fy[j] = iy - yfl;
#endif
                    const int x0 = (int)fminf(fmaxf(xfl, -2.0f), (float)(a.W + 1));
                    const int y0 = (int)fminf(fmaxf(yfl, -2.0f), (float)(a.H + 1));
                    const bool any = live & (k < a.D) & (ix == ix) & (iy == iy) & ((unsigned)(x0 + 1) < (unsigned)(a.W + 1)) &
                                     ((unsigned)(y0 + 1) < (unsigned)(a.H + 1));
                    ki[j] = any ? ((y0 << 16) | (x0 & 0xffff)) : KEY_NONE;
                    if ((j & 3) == 3) __builtin_amdgcn_sched_barrier(0);  // four divide chains at a time
                }

                CELLS_STAMP(1)  // positions
                int jlo = 0;
                while (jlo < nstep) {  // passes of this window (block-uniform)
                    // ---- scan: cell slot of every plane from step jlo on, bounding box, cut point ----
                    int slot[JS];
                    unsigned newmask = 0;   // bit j: the plane of step j opens a new cell
                    int run = 0, jok = jlo, carry = KEY_NONE;
                    const int mge1 = opaque_v(g >= 1 ? -1 : 0), mge2 = opaque_v(g >= 2 ? -1 : 0);  // lane masks as values
                    int bx0 = INT_MAX, by0 = INT_MAX, bx1 = INT_MIN, by1 = INT_MIN;
#pragma unroll
                    for (int j = 0; j < JS; ++j) {
                        slot[j] = 0;
                        if (FAST || (j >= jlo && j < nstep)) {  // uniform
                            const int key = ki[j];
                            const int left = CELLS_DPP_I(key, QP_SHR1);
                            const int prev = (left & mge1) | (carry & ~mge1);
                            const int isn = ((key != KEY_NONE) & (key != prev)) ? 1 : 0;
                            int s1 = CELLS_DPP_I(isn, QP_SHR1);
                            s1 = isn + (s1 & mge1);
                            int s2 = CELLS_DPP_I(s1, QP_SHR2);
                            s2 = s1 + (s2 & mge2);
                            slot[j] = max(run + s2 - 1, 0);
                            run += CELLS_DPP_I(s2, QP_B3);
                            carry = CELLS_DPP_I(key, QP_B3);
                            newmask |= (unsigned)isn << j;
                            if (run <= NS) jok = j + 1;  // every pixel can take at least one step: 4 cells <= NS
                        }
                    }
                    // block-wide: largest number of steps every pixel has slots for
                    auto exchange2 = [&](int v0, int v1, int v2, int v3, int v4, int& o0, int& o1, int& o2, int& o3, int& o4) {
                        // v0: min-reduced, v1..v2: min-reduced, v3..v4: max-reduced
                        const int w0 = wave_min_s(v0), w1 = wave_min_s(v1), w2 = wave_min_s(v2), w3 = wave_max_s(v3), w4 = wave_max_s(v4);
                        int (*sx)[8] = s_ex[ex_parity];
                        ex_parity ^= 1;
                        if (lane == 0) { sx[wave][0] = w0; sx[wave][1] = w1; sx[wave][2] = w2; sx[wave][3] = w3; sx[wave][4] = w4; }
                        __syncthreads();
                        o0 = INT_MAX; o1 = INT_MAX; o2 = INT_MAX; o3 = INT_MIN; o4 = INT_MIN;
#pragma unroll
                        for (int w = 0; w < NW; ++w) {
                            o0 = min(o0, sx[w][0]); o1 = min(o1, sx[w][1]); o2 = min(o2, sx[w][2]);
                            o3 = max(o3, sx[w][3]); o4 = max(o4, sx[w][4]);
                        }
                    };
                    auto bbox_upto = [&](int jhi_) {
                        bx0 = INT_MAX; by0 = INT_MAX; bx1 = INT_MIN; by1 = INT_MIN;
#pragma unroll
                        for (int j = 0; j < JS; ++j) {
                            if ((FAST || (j >= jlo && j < jhi_)) && ki[j] != KEY_NONE) {
                                const int y0 = ki[j] >> 16, x0 = (int)(short)(ki[j] & 0xffff);
                                bx0 = min(bx0, x0); bx1 = max(bx1, x0);
                                by0 = min(by0, y0); by1 = max(by1, y0);
                            }
                        }
                    };
                    // The first exchange carries the cut point and the box of ALL remaining steps; if the cut point
                    // or the window capacity then shortens the pass, the box is reduced again (rare).
                    bbox_upto(nstep);
                    int jhi, wx0, wy0, wx1, wy1;
                    exchange2(jok, bx0, by0, bx1, by1, jhi, wx0, wy0, wx1, wy1);
                    int pitch = 0, WR = 0;
                    bool recompute = jhi < nstep;
                    for (;;) {
                        if (recompute) {
                            bbox_upto(jhi);
                            int dummy;
                            exchange2(0, bx0, by0, bx1, by1, dummy, wx0, wy0, wx1, wy1);
                        }
                        if (wx0 > wx1) { wx0 = 0; wx1 = 0; wy0 = 0; wy1 = 0; }  // no tap of these planes is inside the image
                        // pitch = 8 mod 16 texels: the two rows of a cell land in different bank halves
                        pitch = ((wx1 - wx0 + 2 + 7) & ~15) + 8;
                        WR = wy1 - wy0 + 2;
                        if (pitch * WR <= WT) break;
                        if (jhi - jlo <= 1) { bail = true; break; }
                        jhi = jlo + (jhi - jlo) / 2;
                        recompute = true;
                    }
                    if (FAST && jhi < JS) bail = true;  // more than one pass: leave the tile to the generic instantiation
                    if (bail) break;
                    CELLS_STAMP(2)  // scan + exchange

                    // ---- cell list -> this lane's corner addresses; plane info -------------------------
                    int nc = 0;
#pragma unroll
                    for (int j = 0; j < JS; ++j) {
                        if (FAST || (j >= jlo && j < jhi)) {  // uniform
                            const int key = ki[j];
                            const bool any = key != KEY_NONE;
                            const int y0 = key >> 16, x0 = (int)(short)(key & 0xffff);
                            const int tex = any ? (y0 - wy0) * pitch + (x0 - wx0) : 0;
                            if ((newmask >> j) & 1u) *(lds_iw)(size_t)(clist0 + (q * NS + slot[j]) * 4) = tex;
                            if (any) nc = max(nc, slot[j] + 1);
                            // plane info: window texel (11 bits) | dump byte offset of the cell (<< 11) | any (bit 31)
                            ki[j] = tex | ((slot[j] * 256 + q * 16) << 11) | (any ? (int)0x80000000 : 0);
                        }
                    }
                    nc = max(nc, CELLS_DPP_I(nc, QP_XOR1));
                    nc = max(nc, CELLS_DPP_I(nc, QP_XOR2));   // cells of this pixel in this pass
                    const int wave_nc = max(wave_max_s(nc), 1);  // slot 0 is always computed: planes without a cell read it with weight 0
                    lds_wait();
                    int addrp[NS / 2];  // two 16-bit LDS byte addresses (ring buffer 0) per register
                    {
                        const int corner = lds0 + ((g & 1) + (g >> 1) * pitch) * 16;
#pragma unroll
                        for (int i4 = 0; i4 < NS / 4; ++i4) {
                            const v4i c4 = *(lds_v4i)(size_t)(clist0 + (q * NS + i4 * 4) * 4);
                            const int a0 = (i4 * 4 + 0 < nc ? c4.x : 0) * 16 + corner;
                            const int a1 = (i4 * 4 + 1 < nc ? c4.y : 0) * 16 + corner;
                            const int a2 = (i4 * 4 + 2 < nc ? c4.z : 0) * 16 + corner;
                            const int a3 = (i4 * 4 + 3 < nc ? c4.w : 0) * 16 + corner;
                            addrp[i4 * 2 + 0] = a0 | (a1 << 16);
                            addrp[i4 * 2 + 1] = a2 | (a3 << 16);
                        }
                    }

                    // ---- staging offsets of this thread's window texels --------------------------------
                    int so[SLOTS];
                    {
                        const float rp = __builtin_amdgcn_rcpf((float)pitch);
#pragma unroll
                        for (int sl = 0; sl < SLOTS; ++sl) {
                            const int tl = sl * NT + tid;
                            const int row = (int)(((float)tl + 0.5f) * rp), col = tl - row * pitch;
                            const int gx = wx0 + col, gy = wy0 + row;
                            const bool inb = row < WR && gx >= 0 && gx < a.W && gy >= 0 && gy < a.H;
                            so[sl] = inb ? (gy * a.W + gx) * 16 : 0x7fffffff;
                        }
                    }
                    const int wtex = pitch * WR;
                    const unsigned my_lds = lds0 + wave * 1024;  // this wave's 64 texels of every 256-texel slot
                    const unsigned my_ref = lds0 + REF_OFF + wave * 256;
                    const int ro = p * 4;
                    int nd = 0;  // window DMA instructions of this wave per chunk (wave-uniform)
#pragma unroll
                    for (int sl = 0; sl < SLOTS; ++sl) nd += (sl * NT + wave * 64 < wtex) ? 1 : 0;
                    // One stage = the wave's share of the window of chunk ch + its 64 reference features
                    // (lane (q, g) <- channel 4 ch + g of pixel q; channels beyond C read as 0 like the packed source).
                    auto stage = [&](int bufi, int ch) {
#pragma unroll
                        for (int sl = 0; sl < SLOTS; ++sl)
                            if (sl * NT + wave * 64 < wtex)  // wave-uniform
#ifdef CELLS_ABL_NO_DMA   // ablation build: every chunk re-reads plane 0 (L2 resident), results are wrong
                                dma_b128(src_rsrc, my_lds + bufi * BUF_BYTES + sl * NT * 16, so[sl], 0);
#else
                                dma_b128(src_rsrc, my_lds + bufi * BUF_BYTES + sl * NT * 16, so[sl], ch * HW * 16);
#endif
                        const int c = ch * 4 + g;
                        dma_b32(ref_rsrc, my_ref + bufi * (NW * 256), c < a.C ? ro + c * HW * 4 : 0x7fffffff, 0);
                    };
                    // Every wave is past its previous plane loop (the exchange barrier above): ring and Gram window are free.
                    {   // Gram planes of the window: plane nchunk as float4, .x of plane nchunk + 1 as floats
#pragma unroll
                        for (int sl = 0; sl < SLOTS; ++sl) {
                            if (sl * NT + wave * 64 < wtex) {
                                dma_b128(src_rsrc, my_lds + GRAMA_OFF + sl * NT * 16, so[sl], nchunk * HW * 16);
                                dma_b32(src_rsrc, lds0 + GRAMB_OFF + (sl * NT + wave * 64) * 4, so[sl], (nchunk + 1) * HW * 16);
                            }
                        }
                    }
#pragma unroll
                    for (int st = 0; st < NBUF - 1; ++st)
                        if (st < nchunk) stage(st, st);

                    CELLS_STAMP(3)  // cell list, addresses, staging offsets, first DMA
                    float X[NS];
#pragma unroll
                    for (int i = 0; i < NS; ++i) X[i] = 0.0f;
                    float rr = 0.0f;

                    // ---- channel loop ---------------------------------------------------------------------
                    // Groups of 4 cells, two register sets: the reads of group n + 1 are in flight while the 16 fma of
                    // group n execute (left to itself the compiler issues read, wait, 4 fma per cell).
#define CELLS_AD(i_) (((i_) & 1) ? (pk_[((i_) >> 1) & 1] >> 16) : (pk_[((i_) >> 1) & 1] & 0xffffu))
#define CELLS_LOAD4(T, G4, BUF)                                                                           \
    {                                                                                                     \
        const unsigned pk_[2] = {(unsigned)opaque_v(addrp[2 * (G4)]), (unsigned)opaque_v(addrp[2 * (G4) + 1])}; \
        T[0] = *(lds_v4f)(size_t)(CELLS_AD(0) + (BUF) * BUF_BYTES);                                       \
        T[1] = *(lds_v4f)(size_t)(CELLS_AD(1) + (BUF) * BUF_BYTES);                                       \
        T[2] = *(lds_v4f)(size_t)(CELLS_AD(2) + (BUF) * BUF_BYTES);                                       \
        T[3] = *(lds_v4f)(size_t)(CELLS_AD(3) + (BUF) * BUF_BYTES);                                       \
    }
#define CELLS_FMA4(T, G4)                                                                                 \
    _Pragma("unroll") for (int u = 0; u < 4; ++u) {                                                       \
        float x_ = X[4 * (G4) + u];                                                                       \
        x_ = __builtin_fmaf(T[u].x, rb0, x_); x_ = __builtin_fmaf(T[u].y, rb1, x_);                       \
        x_ = __builtin_fmaf(T[u].z, rb2, x_); x_ = __builtin_fmaf(T[u].w, rb3, x_);                       \
        X[4 * (G4) + u] = x_;                                                                             \
    }
    // step G4: reads of group G4 + SETS - 1 into the set that the fma of group G4 - 1 released, then the fma of group G4
#define CELLS_STEP(G4, BUF)                                                                               \
    if (4 * (G4) < wave_nc) {                                                                             \
        if (4 * ((G4) + SETS - 1) < wave_nc && (G4) + SETS - 1 < NS / 4)                                  \
            CELLS_LOAD4(ts[((G4) + SETS - 1) % SETS], ((G4) + SETS - 1 < NS / 4 ? (G4) + SETS - 1 : 0), BUF) \
        __builtin_amdgcn_sched_barrier(0);                                                                \
        CELLS_FMA4(ts[(G4) % SETS], G4)                                                                   \
        __builtin_amdgcn_sched_barrier(0);                                                                \
    }
#ifdef CELLS_ABL_NO_X   // ablation build: only the first group of 4 cells is accumulated, results are wrong
#define CELLS_ABL_REST(BUF)
#else
#define CELLS_ABL_REST(BUF)                                                                               \
        CELLS_STEP(1, BUF) CELLS_STEP(2, BUF) CELLS_STEP(3, BUF) CELLS_STEP(4, BUF) CELLS_STEP(5, BUF)    \
        CELLS_STEP(6, BUF) CELLS_STEP(7, BUF)
#endif
#define CELLS_CHUNK_HEAD(BUF)                                                                             \
        const float rc = *(lds_f)(size_t)(my_ref + (BUF) * (NW * 256) + lane * 4);                        \
        v4f ts[SETS][4];                                                                                  \
        CELLS_LOAD4(ts[0], 0, BUF)
#define CELLS_CHUNK_R(BUF)                                                                                \
        const float rb0 = CELLS_DPP_F(rc, QP_B0), rb1 = CELLS_DPP_F(rc, QP_B1);                           \
        const float rb2 = CELLS_DPP_F(rc, QP_B2), rb3 = CELLS_DPP_F(rc, QP_B3);                           \
        rr = __builtin_fmaf(rc, rc, rr);
    // generic: every step guarded by the wave's cell count
#define CELLS_CHUNK_GENERIC(BUF)                                                                          \
    {                                                                                                     \
        CELLS_CHUNK_HEAD(BUF)                                                                             \
        if (SETS > 2 && 4 < wave_nc) CELLS_LOAD4(ts[1 % SETS], 1, BUF)                                    \
        if (SETS > 3 && 8 < wave_nc) CELLS_LOAD4(ts[2 % SETS], 2, BUF)                                    \
        CELLS_CHUNK_R(BUF)                                                                                \
        CELLS_STEP(0, BUF) CELLS_ABL_REST(BUF)                                                            \
    }
    // straight line for NG4 groups of 4 cells (slots beyond the wave's cell count point at window texel 0)
#define CELLS_SSTEP(G4, NG4, BUF)                                                                         \
    if ((G4) < (NG4)) {                                                                                   \
        if ((G4) + SETS - 1 < (NG4)) CELLS_LOAD4(ts[((G4) + SETS - 1) % SETS], ((G4) + SETS - 1 < NS / 4 ? (G4) + SETS - 1 : 0), BUF) \
        __builtin_amdgcn_sched_barrier(0);                                                                \
        CELLS_FMA4(ts[(G4) % SETS], G4)                                                                   \
        __builtin_amdgcn_sched_barrier(0);                                                                \
    }
#define CELLS_CHUNK_STRAIGHT(NG4, BUF)                                                                    \
    {                                                                                                     \
        CELLS_CHUNK_HEAD(BUF)                                                                             \
        if (SETS > 2 && 1 < (NG4)) CELLS_LOAD4(ts[1 % SETS], 1, BUF)                                      \
        if (SETS > 3 && 2 < (NG4)) CELLS_LOAD4(ts[2 % SETS], 2, BUF)                                      \
        CELLS_CHUNK_R(BUF)                                                                                \
        CELLS_SSTEP(0, NG4, BUF) CELLS_SSTEP(1, NG4, BUF) CELLS_SSTEP(2, NG4, BUF) CELLS_SSTEP(3, NG4, BUF) \
        CELLS_SSTEP(4, NG4, BUF) CELLS_SSTEP(5, NG4, BUF) CELLS_SSTEP(6, NG4, BUF) CELLS_SSTEP(7, NG4, BUF) \
    }
#define CELLS_CHUNK(BUF)                                                                                  \
    if (!FAST) CELLS_CHUNK_GENERIC(BUF)                                                                   \
    else if (wave_nc <= 8) CELLS_CHUNK_STRAIGHT(2, BUF)                                                   \
    else if (wave_nc <= 16) CELLS_CHUNK_STRAIGHT(4, BUF)                                                  \
    else if (wave_nc <= 24) CELLS_CHUNK_STRAIGHT(6, BUF)                                                  \
    else CELLS_CHUNK_STRAIGHT(8, BUF)
                    // Iteration ch: wait until this wave's stage ch has landed (younger stages stay in flight), barrier
                    // (=> chunk ch is in LDS for every wave, and everybody is done with chunk ch - 1), refill the buffer
                    // of chunk ch - 1 with chunk ch + NBUF - 1, compute chunk ch.  One copy of the compute code per
                    // buffer, so that the buffer base is an immediate of the ds_read.
#define CELLS_ITER(U)                                                                                     \
    if (ch0 + (U) < nchunk) {                                                                             \
        const int ch = ch0 + (U);                                                                         \
        wait_but(min(NBUF - 2, nchunk - 1 - ch) * (nd + 1));                                              \
        CELLS_STAMP(10)                                                                                   \
        lds_barrier();                                                                                    \
        CELLS_STAMP(11)                                                                                   \
        if (ch + NBUF - 1 < nchunk) stage(((U) + NBUF - 1) % NBUF, ch + NBUF - 1);                        \
        CELLS_STAMP(12)                                                                                   \
        CELLS_CHUNK(U)                                                                                    \
        CELLS_STAMP(4)                                                                                    \
    }
                    for (int ch0 = 0; ch0 < nchunk; ch0 += NBUF) {
                        CELLS_ITER(0)
                        CELLS_ITER(1)
                        if (NBUF > 2) { CELLS_ITER(NBUF > 2 ? 2 : 0) }
                    }
#undef CELLS_ITER
#undef CELLS_CHUNK
#undef CELLS_CHUNK_STRAIGHT
#undef CELLS_SSTEP
#undef CELLS_CHUNK_GENERIC
#undef CELLS_CHUNK_R
#undef CELLS_CHUNK_HEAD
#undef CELLS_STEP
#undef CELLS_FMA4
#undef CELLS_LOAD4
#undef CELLS_AD
                    CELLS_STAMP(4)  // channel loop
                    // |r|^2 of the pixel: the quad's four channel residues
                    rr = rr + CELLS_DPP_F(rr, QP_XOR1);
                    rr = rr + CELLS_DPP_F(rr, QP_XOR2);
                    lds_barrier();  // every wave is done with the ring: the dumps may overwrite it
#pragma unroll
                    for (int i = 0; i < NS; ++i)
                        if (i < wave_nc) *(lds_fw)(size_t)(dump0 + i * 256 + lane * 4) = X[i];
                    lds_wait();

                    CELLS_STAMP(5)  // barrier + dump
                    // ---- plane loop: planes kw + 4 j + g of the steps of this pass --------------------------
                    const unsigned gA = lds0 + GRAMA_OFF, gB = lds0 + GRAMB_OFF;
                    const int gk2 = opaque_v(kw + g);
                    // LDS operands of one plane: X of the 4 corners, Gram terms of the cell
                    struct PlaneOps { v4f Xc, G00, G01, G10; float N11, D2; };
                    auto fetch = [&](int info) {
                        const int tex = info & 0x7ff;
                        const unsigned xa = dump0 + (((unsigned)info >> 11) & 0xfffffu);
                        PlaneOps o;
                        o.Xc = *(lds_v4f)(size_t)xa;
                        o.G00 = *(lds_v4f)(size_t)(gA + tex * 16);
                        o.G01 = *(lds_v4f)(size_t)(gA + tex * 16 + 16);
                        o.G10 = *(lds_v4f)(size_t)(gA + (tex + pitch) * 16);
                        o.N11 = *(lds_f)(size_t)(gA + (tex + pitch) * 16 + 16);
                        o.D2 = *(lds_f)(size_t)(gB + tex * 4);
                        return o;
                    };
                    // software pipeline: the reads of step j + 1 are in flight while step j is combined
                    PlaneOps cur = fetch(ki[0]), nxt;
                    if (!FAST) {
#pragma unroll
                        for (int j = 1; j < JS; ++j)
                            if (j == jlo) cur = fetch(ki[j]);  // uniform
                    }
                    nxt = cur;
#pragma unroll
                    for (int j = 0; j < JS; ++j) {
                        if (FAST || (j >= jlo && j < jhi)) {  // uniform
                            if (j + 1 < JS && (FAST || j + 1 < jhi)) nxt = fetch(ki[j + 1 < JS ? j + 1 : j]);
                            __builtin_amdgcn_sched_barrier(0);
                            const bool any = ki[j] < 0;
#if CELLS_KEEP_POS
                            float fw = fx[j], fe = 1.0f - fw, fn = fy[j], fs = 1.0f - fn;
#else
                            // the position is recomputed (its divide chains run in the shadow of the LDS reads): 32
                            // registers less across the channel loop
                            float ix, iy;
                            plane_sample_pos_fast(xf, t2a, t2b, t2c, dtab[min(gk2 + 4 * j, a.D - 1)], cx, cy, rcx, rcy, half_w, half_h, ix, iy);
                            float fw = ix - floorf(ix), fe = 1.0f - fw, fn = iy - floorf(iy), fs = 1.0f - fn;
#endif
                            if (!any) { fw = fw * 0.0f; fe = fe * 0.0f; fn = fn * 0.0f; fs = fs * 0.0f; }  // NaN stays NaN, like ATen
                            // |sum_t w_t s_t|^2, separable in the x weights (e, w) and the y weights (s, n)
                            const float ee = fe * fe, ww = fw * fw, ew = fe * fw;
                            const float A = ee * cur.G00.x + ww * cur.G01.x + 2.0f * ew * cur.G00.y;   // top row:    N00, N01, H00
                            const float B = ee * cur.G10.x + ww * cur.N11 + 2.0f * ew * cur.G10.y;     // bottom row: N10, N11, H10
                            const float Cq = ee * cur.G00.z + ww * cur.G01.z + ew * (cur.G00.w + cur.D2);  // cross rows: V00, V01, D1 + D2
                            const float Q = (fs * fs) * A + (fn * fn) * B + 2.0f * (fs * fn) * Cq;
                            const float XW = (fs * fe) * cur.Xc.x + (fs * fw) * cur.Xc.y + (fn * fe) * cur.Xc.z + (fn * fw) * cur.Xc.w;
                            const float c = div_sigma((Q - 2.0f * XW) + rr);
                            float& o = cost[m * JS + j];
                            o = (!MV || v == 0) ? (0.0f + c) : (o + c);
                            cur = nxt;
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                    jlo = jhi;
                    CELLS_STAMP(6)  // plane loop
                }  // passes
            }      // windows
        }          // views

        if (bail) {
            if (tid == 0) { *my_flag = 2; atomicAdd(&queue[GATHER_COUNT_SLOT - 8], 1); }   // (queue = workspace counters + 8)
        } else {
            // ---- epilogue from registers: cost store, log-softmax over D, expectation --------------------
            const size_t obase = (size_t)b * a.D * HW + p;
            float* const cost_out = CELLS_ARG(float*, cost_out);
            float* const logp_out = CELLS_ARG(float*, logp_out);
            float* const depth_out = CELLS_ARG(float*, depth_out);
            const int ge = opaque_v(g);
            if (cost_out && live) {
#pragma unroll
                for (int i = 0; i < NWIN * JS; ++i) {
                    const int k = (i / JS) * 4 * JS + (i % JS) * 4 + ge;
                    if (k < a.D) cost_out[obase + (size_t)k * HW] = cost[i];
                }
            }
            if (logp_out || depth_out) {
                float mx = -INFINITY;
#pragma unroll
                for (int i = 0; i < NWIN * JS; ++i) {
                    const int k = (i / JS) * 4 * JS + (i % JS) * 4 + ge;
                    if (k < a.D) mx = fmaxf(mx, cost[i]);
                }
                mx = fmaxf(mx, CELLS_DPP_F(mx, QP_XOR1));
                mx = fmaxf(mx, CELLS_DPP_F(mx, QP_XOR2));
                // p_k = e_k / s with e_k = exp(c_k - max): one exp per plane; log p_k = (c_k - max) - log s
                float ssum = 0.0f, esum = 0.0f;
#pragma unroll
                for (int i = 0; i < NWIN * JS; ++i) {
                    const int k = (i / JS) * 4 * JS + (i % JS) * 4 + ge;
                    if (k < a.D) {
                        const float ek = expf(cost[i] - mx);
                        ssum = ssum + ek;
                        esum = __builtin_fmaf(dtab[k], ek, esum);
                    }
                    if ((i & 3) == 3) __builtin_amdgcn_sched_barrier(0);
                }
                ssum = ssum + CELLS_DPP_F(ssum, QP_XOR1);
                ssum = ssum + CELLS_DPP_F(ssum, QP_XOR2);
                esum = esum + CELLS_DPP_F(esum, QP_XOR1);
                esum = esum + CELLS_DPP_F(esum, QP_XOR2);
                const float ls = logf(ssum);
                if (logp_out && live) {
#pragma unroll
                    for (int i = 0; i < NWIN * JS; ++i) {
                        const int k = (i / JS) * 4 * JS + (i % JS) * 4 + ge;
                        if (k < a.D) logp_out[obase + (size_t)k * HW] = (cost[i] - mx) - ls;
                    }
                }
                const float e = esum / ssum;
                if (depth_out && live && g == 0) depth_out[(size_t)b * HW + p] = e;
            }
        }
        CELLS_STAMP(7)  // epilogue
        __syncthreads();  // s_item of the next round is visible; nobody still reads this tile's LDS state
        item_par ^= 1;
        item = s_item[item_par];
        CELLS_STAMP(8)  // end-of-tile barrier
    }
#ifdef CELLS_STAMPS
    if (lane == 0)
        for (int i = 0; i < 13; ++i) atomicAdd(reinterpret_cast<unsigned long long*>(queue + 16) + i, stamp_acc[i]);
#endif
}

// ---- host side -------------------------------------------------------------------------------------------

namespace {

struct DeviceInfo {
    int n_cu = 0;
    bool lds_raised[8] = {false, false, false, false, false, false, false, false};
};
DeviceInfo& device_info(int dev) {
    static DeviceInfo info[64];
    return info[dev < 0 || dev >= 64 ? 0 : dev];
}

size_t cells_lds_bytes(int D) { return (size_t)DTAB_OFF + (size_t)((D + 3) & ~3) * sizeof(float); }

}  // namespace

int sweep_cells_max_planes() { return 128; }

// Launches the pre-pass, the cell-list kernel, then the gather kernel on the tiles it flagged.
// The workspace layout is the one of sweep_tiled.hip (flags, queue counters, packed source).
hipError_t launch_sweep_cells(const SweepArgs& a, void* workspace, hipStream_t stream, bool packed_ready) {
    const int tiles_x = (a.W + TW - 1) / TW, tiles_y = (a.H + TH - 1) / TH;
    const int tiles = tiles_x * tiles_y;
    const size_t flag_only = ((size_t)a.B * tiles * sizeof(int) + 255) & ~(size_t)255;
    int* flags = reinterpret_cast<int*>(workspace);
    int* queue = reinterpret_cast<int*>(static_cast<char*>(workspace) + flag_only);
    float4* packed = reinterpret_cast<float4*>(static_cast<char*>(workspace) + flag_only + 256);
    hipError_t e = packed_ready ? clear_sweep_flags(a, workspace, stream) : launch_pack_c4(a, workspace, stream, /*centre=*/false);
    if (e != hipSuccess) return e;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    DeviceInfo& di = device_info(dev);
    di.n_cu = sweep_device_cus();
    const size_t lds = cells_lds_bytes(a.D);
    int nblk = (di.n_cu * CELLS_OCC + 7) & ~7;  // persistent grid: CELLS_OCC blocks per CU, a multiple of 8
    const long long full = 8ll * ((tiles + 7) / 8) * a.B;
    if (full <= nblk) nblk = (int)full;
    const int nwin = a.D <= 64 ? 1 : 2, mv = a.V > 1 ? 1 : 0;
    typedef void (*kern_t)(SweepArgs, const float4*, int*, int*, int, int, const int*);
    static const kern_t kerns[4] = {sweep_cells_kernel<1, false, false>, sweep_cells_kernel<1, true, false>,
                                    sweep_cells_kernel<2, false, false>, sweep_cells_kernel<2, true, false>};
    const int which = (nwin - 1) * 2 + mv;
    if (!di.lds_raised[which]) {
        e = hipFuncSetAttribute((const void*)kerns[which], hipFuncAttributeMaxDynamicSharedMemorySize, (int)cells_lds_bytes(128));
        if (e != hipSuccess) return e;
        di.lds_raised[which] = true;
    }
    // Straight-line kernel first (sweep_cells_fast.hip, D = 64 or 128 only); it appends the tiles it cannot do to
    // redo_list (count in queue[8]), which then is the work list of the generic kernel.  Otherwise the generic kernel
    // takes every tile.
    int* redo_list = reinterpret_cast<int*>(reinterpret_cast<char*>(packed) + (size_t)a.B * a.V * ((a.C + 3) / 4 + 2) * a.H * a.W * sizeof(float4));
    const bool fast_ok = a.D == 64 * nwin;
    if (fast_ok) {
        e = launch_sweep_cells_fast(a, packed, flags, queue, redo_list, tiles_x, tiles, di.n_cu, stream);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kerns[which], dim3(nblk), dim3(NT), lds, stream, a, packed, flags, queue + 8, tiles_x, tiles,
                       fast_ok ? (const int*)redo_list : (const int*)nullptr);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    SweepArgs ag = a;
    ag.packed_src = packed;   // (the gather kernel's source when the caller passed a packed source only)
    return launch_sweep_direct_flagged(ag, flags, queue + GATHER_COUNT_SLOT, tiles_x, tiles, stream, 2);
}

}  // namespace pdepth
