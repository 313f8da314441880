// Cell-list plane sweep, straight-line instantiation (L2 metric, D = 64 or 128): the kernel the benchmark runs.
//
// Same algorithm as sweep_cells.hip (see its header: per-pixel cell lists, correlation form
// sum_c (sum_t w_t s_t[c] - r[c])^2 = w^T G w - 2 sum_t w_t X_t + |r|^2, four lanes per reference pixel), replacing
// est_swp_volume_v4 / _back_warp_homo_parallel / img_dis_L2_pard (warping/homography.py:98-135, :170-198, :80-82) and
// the log_softmax + dpv_to_depthmap tail (models/packnet.py:394, utils/img_utils.py:52-61) -- specialised so that
// every hot loop is straight-line code (hipcc drains lgkmcnt at every control-flow join, which serialises
// "read, wait, compute" through guarded code) and the channel loop touches every source texel of a pixel once:
//
//   roles   = lane r of a pixel's quad owns the texels with (x & 1, y & 1) == (r & 1, r >> 1).  Every bilinear cell
//             has exactly one texel of each parity class, and a texel shared by consecutive cells of the epipolar
//             walk keeps its owner, so "same texel as in my previous cell" is an in-lane comparison: the channel
//             loop accumulates X once per distinct texel (13 per lane on the benchmark poses, against 24 cells).
//   planes  = lane r owns planes 16 r .. 16 r + 15 of the 64-plane window (contiguous): the cell scan -- which
//             planes open a new cell, which of its four texels are new -- runs inside the lane; the only exchange
//             is one quad prefix sum of the four new-texel counters, packed in one register.
//   lists   = a plane lane writes the cell key into the slot of each role (texel lists in LDS); the role lanes
//             read their 16 slots back and derive their own texel of each cell from the parity.
//   X dump  = [slot][pixel][role] floats, 4 KB per wave; a plane reads its four X by the four slot numbers packed in
//             its info word (five bits per role), next to the window texel of its cell and the cell parity.
//   ref     = one LDS-DMA per wave and FOUR chunks: 16 channels x 16 pixels in one buffer_load_dwordx4 ... lds.
//
// One pass per window and view, D = 64 * NWIN exactly; a tile that needs more than 16 texel slots per role or more
// than WT window texels is flagged 1 and redone by the generic kernel (sweep_cells.hip), which hands what it cannot do
// to the gather kernel.  2 blocks of 4 waves per CU (256 VGPRs per lane: positions, infos and accumulators stay in
// registers), 79 KB of LDS per block.
#include <hip/hip_runtime.h>

#include "cells_common.hpp"
#include "../geometry.hpp"
#include "../kernels.hpp"

namespace pdepth {

namespace {

namespace fast {
constexpr int TW = 16, TH = 4;
constexpr int NW = 4, NT = 64 * NW;
constexpr int JS = 16;           // planes per lane and window
constexpr int NTS = 24;          // texel slots per role and pass (5-bit slot fields: at most 31)
#ifndef FCELLS_WT
#define FCELLS_WT 1024
#endif
#ifndef FCELLS_PREFETCH   // plane loop: LDS operands of plane j + 1 in flight during plane j
#define FCELLS_PREFETCH 0
#endif
#ifndef FCELLS_SETS
#define FCELLS_SETS 3
#endif
constexpr int WT = FCELLS_WT;    // window texels per ring buffer
constexpr int NBUF = 4;          // window ring depth: chunks ch + 1 .. ch + 3 are in flight while chunk ch is computed
constexpr int SETS = FCELLS_SETS;  // register sets of 4 texels in the channel loop: SETS - 1 groups of reads in flight
constexpr int SLOTS = (WT + NT - 1) / NT;
constexpr int BUF_BYTES = WT * 16;
constexpr int RING_BYTES = NBUF * BUF_BYTES;
constexpr int DUMP_WAVE_BYTES = NTS * 256;      // X dump of a wave [slot][16 px][4 roles]
constexpr int TLIST_WAVE_BYTES = 16 * 4 * NTS * 2;  // texel lists of a wave [16 px][4 roles][NTS] 16-bit entries
// The two Gram planes of the window ride the ring as "chunks" nchunk (float4) and nchunk + 1 (floats): they are staged
// into the buffers that free up during the last iterations of the channel loop and stay there for the plane loop; the
// X dumps take the two buffers left.
constexpr int REF_OFF = RING_BYTES;             // [2 stages][NW][16 channels][16 px] floats
constexpr int DTAB_OFF = REF_OFF + 2 * NW * 1024;
static_assert(NTS == 24, "the channel loop is written out for 6 groups of 4 slots");
static_assert(NTS % 8 == 0 && NTS <= 31, "texel lists are read back 8 entries at a time; 5-bit slot fields");
static_assert(WT % 64 == 0 && SLOTS <= 4, "whole DMA instructions, at most 4 per chunk and wave: 9 in flight (wait_but)");
static_assert(NBUF == 4 && 2 * DUMP_WAVE_BYTES <= BUF_BYTES && NW * TLIST_WAVE_BYTES <= BUF_BYTES, "dumps (two waves per buffer) / texel lists alias the ring");
static_assert(BUF_BYTES + 1024 <= 65536 && (NBUF - 1) * BUF_BYTES <= 65535, "packed 16-bit LDS addresses of the window taps (buffer 0) + immediate buffer offset");
static_assert(2 * (DTAB_OFF + 512 + 320) <= 160 * 1024, "two blocks per CU");
static_assert(WT <= 1024, "window texel index: 10 bits of the plane info, 1023 = no tap inside the image (windows are kept below WT)");
constexpr int F1 = 0x8421;       // one in each of the four 5-bit fields (role r at bits 5 r .. 5 r + 4)
constexpr int KEY_NONE = INT_MIN;
}  // namespace fast

}  // namespace

template <int NWIN, bool MV>
__global__ __launch_bounds__(fast::NT, 2) void sweep_cells_fast_kernel(SweepArgs a, const float4* __restrict__ packed,
                                                                   int* __restrict__ tile_flags, int* __restrict__ queue,
                                                                   int* __restrict__ redo_list, int tiles_x, int ntile) {
    using namespace fast;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    __shared__ int s_ex[2][NW][8];
    __shared__ int s_item[2];
    int ex_parity = 0;

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int q = lane >> 2;   // pixel of the wave's row
    const int r = lane & 3;    // role: texel parity class (x & 1) + 2 (y & 1); planes 16 r .. 16 r + 15; channel 4 ch + r
    const unsigned lds0 = lds_addr_of(lds_raw);
    float* dtab = reinterpret_cast<float*>(lds_raw + DTAB_OFF);
    // texel lists: in the last ring buffer, dead between the block-wide exchange of a pass and the barrier of chunk 0
    const unsigned tlist0 = lds0 + (NBUF - 1) * BUF_BYTES + wave * TLIST_WAVE_BYTES + q * (4 * NTS * 2);

    const int HW = a.H * a.W;
    const int nchunk = (a.C + 3) / 4, nstage = (nchunk + 3) / 4;
    const float half_w = (float)a.W / 2.0f, half_h = (float)a.H / 2.0f;
    const float sigma = a.sigma, rsigma = refined_rcp(sigma);
    auto div_sigma = [&](float v) {
        float o = div_core(v, sigma, rsigma);
        if (!(fabsf(v) < 1.0e30f)) o = v / sigma;  // inf / NaN exactly like the IEEE divide
        return o;
    };
    for (int k = tid; k < a.D; k += NT) dtab[k] = CELLS_ARG(const float*, d_candi)[k];
    __syncthreads();

    // ---- persistent work loop (XCD-aware item order, see sweep_tiled.hip) ---------------------------------
    const int xcd = blockIdx.x & 7, qq = ntile >> 3, rr8 = ntile & 7;
    const int band_first = xcd < rr8 ? xcd * (qq + 1) : rr8 * (qq + 1) + (xcd - rr8) * qq;
    const int band_tiles = qq + (xcd < rr8 ? 1 : 0);
    const int nitems = band_tiles * a.B;
    const bool colmajor = rr8 == 0 && qq % tiles_x == 0;
    const bool queued = (int)gridDim.x < 8 * ((ntile + 7) / 8) * a.B;
    if (queued) {
        if (tid == 0) s_item[0] = atomicAdd(&queue[xcd], 1);
        __syncthreads();
    }
    int item = queued ? s_item[0] : (int)(blockIdx.x >> 3), item_par = 0;
#ifdef CELLS_STAMPS
    unsigned long long stamp_acc[13] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long stamp_t = __builtin_amdgcn_s_memtime();
#endif
    while (item < nitems) {
        CELLS_STAMP(9)
        if (tid == 0) s_item[item_par ^ 1] = queued ? atomicAdd(&queue[xcd], 1) : nitems;
        const int b = item / band_tiles;
        int tile;
        {
            const int ti = item - b * band_tiles;
            tile = band_first + ti;
            if (colmajor) {
                const int band_rows = qq / tiles_x;
                tile = (xcd * band_rows + ti % band_rows) * tiles_x + ti / band_rows;
            }
        }
        const int px = (tile % tiles_x) * TW + q, py = (tile / tiles_x) * TH + wave;
        const bool live = px < a.W && py < a.H;
        const int p = min(py, a.H - 1) * a.W + min(px, a.W - 1);

        const float* cxcy_ = CELLS_ARG(const float*, cxcy);
        const float cx = cxcy_[b * 2 + 0], cy = cxcy_[b * 2 + 1];
        const float rcx = refined_rcp(cx), rcy = refined_rcp(cy);
        const float* rays_ = CELLS_ARG(const float*, rays) + (size_t)b * 3 * HW + p;
        const float r0 = rays_[0], r1 = rays_[HW], r2 = rays_[2 * (size_t)HW];
        CELLS_STAMP(0)  // tile start

        float cost[NWIN * JS];  // cost of plane 64 m + 16 r + j at [m * JS + j]
        bool bail = false;      // block-uniform: the tile goes to the generic kernel

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
                if (bail) break;
                const int kb = opaque_v(m * 4 * JS + JS * r);  // first plane of this lane

                // ---- positions of the lane's 16 planes: cell key, fractional position -----------------------
                float fx[JS], fy[JS];
                int ki[JS];  // cell key (y0 << 16 | x0 & 0xffff) or KEY_NONE; later the packed plane info
                int bx0 = INT_MAX, by0 = INT_MAX, bx1 = INT_MIN, by1 = INT_MIN;
                {
                    v4f d4[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) d4[i] = *(lds_v4f)(size_t)(lds0 + DTAB_OFF + (kb + 4 * i) * 4);
#pragma unroll
                    for (int j = 0; j < JS; ++j) {
                        float ix, iy;
                        plane_sample_pos_fast(xf, t2a, t2b, t2c, d4[j >> 2][j & 3], cx, cy, rcx, rcy, half_w, half_h, ix, iy);
                        const float xfl = floorf(ix), yfl = floorf(iy);
                        fx[j] = ix - xfl;
                        fy[j] = iy - yfl;
                        const int x0 = (int)fminf(fmaxf(xfl, -2.0f), (float)(a.W + 1));
                        const int y0 = (int)fminf(fmaxf(yfl, -2.0f), (float)(a.H + 1));
                        const bool any = live & (ix == ix) & (iy == iy) & ((unsigned)(x0 + 1) < (unsigned)(a.W + 1)) &
                                         ((unsigned)(y0 + 1) < (unsigned)(a.H + 1));
                        ki[j] = any ? ((y0 << 16) | (x0 & 0xffff)) : KEY_NONE;
                        bx0 = any ? min(bx0, x0) : bx0; bx1 = any ? max(bx1, x0) : bx1;
                        by0 = any ? min(by0, y0) : by0; by1 = any ? max(by1, y0) : by1;
                        // materialise the three values here: left alone, the compiler keeps ix, floor(ix), ... alive
                        // instead and sinks the subtraction (and, below, whole plane evaluations) to the first use
                        asm volatile("" : "+v"(fx[j]), "+v"(fy[j]), "+v"(ki[j]));
                        if ((j & 3) == 3) __builtin_amdgcn_sched_barrier(0);  // four divide chains at a time
                    }
                }
                CELLS_STAMP(1)  // positions

                // ---- scan, pass 1: which planes open a cell, which of its four texels are new -------------
                // E / O = the even / odd one of the cell's two columns (rows): the texel of role (xpar, ypar) is
                // (xpar ? Ox : Ex, ypar ? Oy : Ey).  A role's texel is new unless it equals the role's texel of the
                // previous cell.  KEY_NONE decodes to coordinates no real cell has, so it never matches.
                int nf[JS];   // new-texel bits of the plane, one per 5-bit field
                int cnt = 0;  // new texels of this lane per role (5-bit fields)
                {
                    int pk = CELLS_DPP_I(ki[JS - 1], QP_SHR1);   // last plane of the previous lane of the quad
                    pk = r == 0 ? KEY_NONE : pk;
                    int pex, pox, pey, poy;
                    {
                        const int x0 = (int)(short)(pk & 0xffff), y0 = pk >> 16;
                        pex = (x0 + 1) & ~1; pox = x0 | 1; pey = (y0 + 1) & ~1; poy = y0 | 1;
                    }
#pragma unroll
                    for (int j = 0; j < JS; ++j) {
                        const int key = ki[j];
                        const int x0 = (int)(short)(key & 0xffff), y0 = key >> 16;
                        const int ex = (x0 + 1) & ~1, ox = x0 | 1, ey = (y0 + 1) & ~1, oy = y0 | 1;
                        const bool valid = key != KEY_NONE;
                        const bool newc = valid & (key != pk);
                        const bool sx0 = ex == pex, sx1 = ox == pox, sy0 = ey == pey, sy1 = oy == poy;
                        int f = ((sx0 & sy0) ? 0 : 1) | ((sx1 & sy0) ? 0 : (1 << 5)) | ((sx0 & sy1) ? 0 : (1 << 10)) |
                                ((sx1 & sy1) ? 0 : (1 << 15));
                        f = newc ? f : 0;
                        nf[j] = f;
                        asm volatile("" : "+v"(nf[j]));
                        cnt += f;
                        // an invalid plane leaves the previous cell in place: the next valid plane compares with it
                        pk = valid ? key : pk;
                        pex = valid ? ex : pex; pox = valid ? ox : pox; pey = valid ? ey : pey; poy = valid ? oy : poy;
                    }
                }
                // quad prefix of the packed counters: slots of this lane's new texels start at `excl`
                int incl = cnt + (CELLS_DPP_I(cnt, QP_SHR1) & (r >= 1 ? -1 : 0));
                incl = incl + (CELLS_DPP_I(incl, QP_SHR2) & (r >= 2 ? -1 : 0));
                const int excl = incl - cnt;
                const int total = CELLS_DPP_I(incl, QP_B3);   // texels per role of this pixel
                // The 5-bit fields hold a lane's own counts (at most JS = 16) but not every quad total (up to 64, e.g. depth
                // candidates in no particular order: every plane a new cell): the totals that decide whether the tile
                // fits are summed in 16-bit fields, and a tile that does not fit never looks at the packed ones.
                int tmax;
                {
                    int w01 = (cnt & 31) | (((cnt >> 5) & 31) << 16), w23 = ((cnt >> 10) & 31) | (((cnt >> 15) & 31) << 16);
                    w01 += CELLS_DPP_I(w01, QP_XOR1); w23 += CELLS_DPP_I(w23, QP_XOR1);
                    w01 += CELLS_DPP_I(w01, QP_XOR2); w23 += CELLS_DPP_I(w23, QP_XOR2);
                    tmax = max(max(w01 & 0xffff, w01 >> 16), max(w23 & 0xffff, w23 >> 16));
                }
                const int mytot = (total >> (5 * r)) & 31;    // texels of this lane's role

                // ---- block-wide: window of all cells; does everything fit one pass? --------------------------
                int wx0, wy0, wx1, wy1, wave_nt;
                {
                    const int w0 = wave_min_s(bx0), w1 = wave_min_s(by0), w2 = wave_max_s(bx1), w3 = wave_max_s(by1);
                    wave_nt = wave_max_s(tmax);
                    int (*sx)[8] = s_ex[ex_parity];
                    ex_parity ^= 1;
                    if (lane == 0) { sx[wave][0] = w0; sx[wave][1] = w1; sx[wave][2] = w2; sx[wave][3] = w3; sx[wave][4] = wave_nt; }
                    __syncthreads();
                    wx0 = INT_MAX; wy0 = INT_MAX; wx1 = INT_MIN; wy1 = INT_MIN;
                    int ntm = 0;
#pragma unroll
                    for (int w = 0; w < NW; ++w) {
                        wx0 = min(wx0, sx[w][0]); wy0 = min(wy0, sx[w][1]);
                        wx1 = max(wx1, sx[w][2]); wy1 = max(wy1, sx[w][3]);
                        ntm = max(ntm, sx[w][4]);
                    }
                    if (ntm > NTS) bail = true;
                }
                if (wx0 > wx1) { wx0 = 0; wx1 = 0; wy0 = 0; wy1 = 0; }  // no tap of this window is inside the image
                // pitch = 8 mod 16 texels: the two rows of a cell land in different bank halves
                const int pitch = ((wx1 - wx0 + 2 + 7) & ~15) + 8;
                const int WR = wy1 - wy0 + 2;
                if (pitch * WR >= WT) bail = true;  // (texel index 1023 means "none" in the plane info)
                if (bail) break;
                CELLS_STAMP(2)  // scan + exchange

                // ---- scan, pass 2: slots of every cell's four texels; texel lists; plane infos -------------------
                {
                    int run = excl;        // texels assigned so far, per role
                    int cs = (excl - F1) & 0xfffff;  // current slot per role: the previous lane's last cell continues here (fields of roles without a texel yet are garbage and replaced before use)
                    const int c0 = -(wy0 * pitch + wx0);
#pragma unroll
                    for (int j = 0; j < JS; ++j) {
                        const int key = ki[j];
                        const bool valid = key != KEY_NONE;
                        const int f = nf[j];
                        run += f;
                        const int msk = f * 31;   // all five bits of every field that got a new texel
                        cs = (msk & (run - F1)) | (~msk & cs);
                        const int x0 = (int)(short)(key & 0xffff), y0 = key >> 16;
                        const int tex = y0 * pitch + x0 + c0;
                        // list entry: window texel of the cell | (x0 & 1) << 10 | (y0 & 1) << 11
                        const int ent = tex | ((x0 & 1) << 10) | ((y0 & 1) << 11);
                        if (valid) {
                            // every role's current slot gets the cell: a role whose texel is shared with the previous
                            // cell derives the same texel from either one
                            *(lds_sw)(size_t)(tlist0 + (0 * NTS + (cs & 31)) * 2) = (short)ent;
                            *(lds_sw)(size_t)(tlist0 + (1 * NTS + ((cs >> 5) & 31)) * 2) = (short)ent;
                            *(lds_sw)(size_t)(tlist0 + (2 * NTS + ((cs >> 10) & 31)) * 2) = (short)ent;
                            *(lds_sw)(size_t)(tlist0 + (3 * NTS + ((cs >> 15) & 31)) * 2) = (short)ent;
                        }
                        // plane info: 4 x 5-bit slot | window texel of the cell << 20 (1023: none) | (x0 & 1, y0 & 1) << 30
                        ki[j] = valid ? (cs | (ent << 20)) : (1023 << 20);
                        asm volatile("" : "+v"(ki[j]));
                    }
                }
                lds_wait();
                // ---- role lanes: texel addresses of the lane's slots --------------------------------------------
                const int wave_n4 = (wave_nt + 3) >> 2;   // groups of 4 slots the wave needs (1..4)
                int addr[NTS];  // LDS byte address of the texel of every slot in ring buffer 0 (the buffer is an immediate of the read)
                {
                    const int xp = r & 1, yp = r >> 1;
#pragma unroll
                    for (int i8 = 0; i8 < NTS / 8; ++i8) {
                        const v4i e4 = *(lds_v4i)(size_t)(tlist0 + (r * NTS + i8 * 8) * 2);
#pragma unroll
                        for (int u = 0; u < 8; ++u) {
                            const int e = (e4[u >> 1] >> (16 * (u & 1))) & 0xffff;
                            // this role's texel of the cell: the column (row) of its parity
                            const int tex = (e & 1023) + (((e >> 10) ^ xp) & 1) + ((((e >> 11) ^ yp) & 1) ? pitch : 0);
                            addr[i8 * 8 + u] = lds0 + (i8 * 8 + u < mytot ? tex : 0) * 16;
                        }
                    }
                }

                // ---- staging offsets of this thread's window texels ------------------------------------------------
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
                const unsigned my_lds = lds0 + wave * 1024;
                const unsigned my_ref = lds0 + REF_OFF + wave * 1024;
                int nd = 0;
#pragma unroll
                for (int sl = 0; sl < SLOTS; ++sl) nd += (sl * NT + wave * 64 < wtex) ? 1 : 0;
                // chunk ch < nchunk: channel group ch; chunk nchunk: Gram plane A (float4); chunk nchunk + 1: .x of Gram
                // plane B as floats.  nd DMA instructions per wave either way.
                auto stage = [&](int bufi, int ch) {
#pragma unroll
                    for (int sl = 0; sl < SLOTS; ++sl) {
                        if (sl * NT + wave * 64 < wtex) {  // wave-uniform
#ifdef CELLS_ABL_NO_DMA   // ablation build: every chunk re-reads plane 0 (L2 resident), results are wrong
                            if (ch <= nchunk) dma_b128(src_rsrc, my_lds + bufi * BUF_BYTES + sl * NT * 16, so[sl], 0);
#else
                            if (ch <= nchunk) dma_b128(src_rsrc, my_lds + bufi * BUF_BYTES + sl * NT * 16, so[sl], ch * HW * 16);
#endif
                            else dma_b32(src_rsrc, lds0 + bufi * BUF_BYTES + (sl * NT + wave * 64) * 4, so[sl], ch * HW * 16);
                        }
                    }
                };
                const int nvirt = nchunk + 2;   // chunks to stage, the two Gram planes included
                // reference features of 16 channels: lane l <- pixels 4 (l & 3) .. + 3 of channel 16 st + (l >> 2)
                // (channels beyond C read as 0 like the packed source; pixels beyond the right image border read the
                //  start of the next row or, past the last channel plane, 0: they belong to dead pixels and are never stored)
                const int ref_voff = (min(py, a.H - 1) * a.W + (tile % tiles_x) * TW + 4 * (lane & 3)) * 4;
                auto stage_ref = [&](int st) {
                    const int c = st * 16 + (lane >> 2);
                    dma_b128(ref_rsrc, my_ref + (st & 1) * (NW * 1024), c < a.C ? ref_voff + c * HW * 4 : 0x7fffffff, 0);
                };
                stage_ref(0);
                stage(0, 0);
                stage(1, 1);   // (nvirt >= 3)
                stage(2, 2);
                CELLS_STAMP(3)  // lists, addresses, staging offsets, first DMA

                float X[NTS];
#pragma unroll
                for (int i = 0; i < NTS; ++i) X[i] = 0.0f;
                float rr = 0.0f;

                // ---- channel loop ---------------------------------------------------------------------------------
#define FC_LOAD4(T, G4, BUF)                                                                              \
    {                                                                                                     \
        T[0] = *(lds_v4f)(size_t)((unsigned)addr[4 * (G4) + 0] + (BUF) * BUF_BYTES);                      \
        T[1] = *(lds_v4f)(size_t)((unsigned)addr[4 * (G4) + 1] + (BUF) * BUF_BYTES);                      \
        T[2] = *(lds_v4f)(size_t)((unsigned)addr[4 * (G4) + 2] + (BUF) * BUF_BYTES);                      \
        T[3] = *(lds_v4f)(size_t)((unsigned)addr[4 * (G4) + 3] + (BUF) * BUF_BYTES);                      \
    }
#define FC_FMA4(T, G4)                                                                                    \
    _Pragma("unroll") for (int u = 0; u < 4; ++u) {                                                       \
        float x_ = X[4 * (G4) + u];                                                                       \
        x_ = __builtin_fmaf(T[u].x, rb0, x_); x_ = __builtin_fmaf(T[u].y, rb1, x_);                       \
        x_ = __builtin_fmaf(T[u].z, rb2, x_); x_ = __builtin_fmaf(T[u].w, rb3, x_);                       \
        X[4 * (G4) + u] = x_;                                                                             \
    }
#define FC_STEP(G4, NG4, BUF)                                                                             \
    if ((G4) < (NG4)) {                                                                                   \
        if ((G4) + SETS - 1 < (NG4)) FC_LOAD4(ts[((G4) + SETS - 1) % SETS], ((G4) + SETS - 1 < NTS / 4 ? (G4) + SETS - 1 : 0), BUF) \
        __builtin_amdgcn_sched_barrier(0);                                                                \
        FC_FMA4(ts[(G4) % SETS], G4)                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                                \
    }
    // straight line for NG4 groups of 4 slots (slots beyond the lane's count point at window texel 0)
#define FC_CHUNK_N(NG4, BUF, CH)                                                                          \
    {                                                                                                     \
        const float rc = *(lds_f)(size_t)(my_ref + (((CH) >> 2) & 1) * (NW * 1024) + ((4 * ((CH) & 3) + r) * 16 + q) * 4); \
        v4f ts[SETS][4];                                                                                  \
        FC_LOAD4(ts[0], 0, BUF)                                                                           \
        if (SETS > 2 && 1 < (NG4)) FC_LOAD4(ts[1 % SETS], 1, BUF)                                         \
        if (SETS > 3 && 2 < (NG4)) FC_LOAD4(ts[2 % SETS], 2, BUF)                                         \
        const float rb0 = CELLS_DPP_F(rc, QP_B0), rb1 = CELLS_DPP_F(rc, QP_B1);                           \
        const float rb2 = CELLS_DPP_F(rc, QP_B2), rb3 = CELLS_DPP_F(rc, QP_B3);                           \
        rr = __builtin_fmaf(rc, rc, rr);                                                                  \
        FC_STEP(0, NG4, BUF) FC_STEP(1, NG4, BUF) FC_STEP(2, NG4, BUF) FC_STEP(3, NG4, BUF)               \
        FC_STEP(4, NG4, BUF) FC_STEP(5, NG4, BUF)                                                         \
    }
#define FC_CHUNK(BUF, CH)                                                                                 \
    if (wave_n4 <= 3) { if (wave_n4 <= 2) FC_CHUNK_N(2, BUF, CH) else FC_CHUNK_N(3, BUF, CH) }            \
    else if (wave_n4 == 4) FC_CHUNK_N(4, BUF, CH)                                                         \
    else { if (wave_n4 == 5) FC_CHUNK_N(5, BUF, CH) else FC_CHUNK_N(6, BUF, CH) }
                // Iteration ch: wait until this wave's window DMA of chunk ch has landed (what iteration ch - 1 issued
                // stays in flight), barrier, refill the buffer of chunk ch - 1 with chunk ch + 2 (and, every fourth
                // chunk, stage the next 16 reference channels), compute chunk ch.
#ifdef CELLS_ABL_NO_WAIT   // ablation build: no DMA wait in the channel loop, results are wrong
#define CELLS_ABL_WAIT(n) (void)(n)
#else
#define CELLS_ABL_WAIT(n) wait_but(n)
#endif
#define FC_ITER(U)                                                                                        \
    if (ch0 + (U) < nchunk) {                                                                             \
        const int ch = ch0 + (U);                                                                         \
        /* in flight behind chunk ch: chunks ch + 1, ch + 2 (if they exist) and the ref stage issued with them */ \
        const int younger = nd * (min(ch + 2, nvirt - 1) - ch) +                                          \
                            ((((ch - 1) & 3) == 0 && ch >= 1 && ((ch - 1) >> 2) + 1 < nstage) ? 1 : 0) +   \
                            ((((ch - 2) & 3) == 0 && ch >= 2 && ((ch - 2) >> 2) + 1 < nstage) ? 1 : 0);    \
        CELLS_ABL_WAIT(younger);                                                                          \
        CELLS_STAMP(10)                                                                                   \
        lds_barrier();                                                                                    \
        CELLS_STAMP(11)                                                                                   \
        if ((ch & 3) == 0 && (ch >> 2) + 1 < nstage) stage_ref((ch >> 2) + 1);                            \
        if (ch + 3 < nvirt) stage(((U) + 3) % NBUF, ch + 3);                                              \
        CELLS_STAMP(12)                                                                                   \
        FC_CHUNK(U, ch)                                                                                   \
        CELLS_STAMP(4)                                                                                    \
    }
                for (int ch0 = 0; ch0 < nchunk; ch0 += NBUF) {
                    FC_ITER(0)
                    FC_ITER(1)
                    FC_ITER(2)
                    FC_ITER(3)
                }
#undef FC_ITER
#undef FC_CHUNK
#undef FC_CHUNK_N
#undef FC_STEP
#undef FC_FMA4
#undef FC_LOAD4
                rr = rr + CELLS_DPP_F(rr, QP_XOR1);
                rr = rr + CELLS_DPP_F(rr, QP_XOR2);
                wait_but(0);    // this wave's share of the Gram planes has landed
                lds_barrier();  // ... everybody's has, and every wave is done with the channel chunks
                // Gram plane A sits in buffer nchunk % 4, B in (nchunk + 1) % 4; the dumps of waves 0, 1 go to buffer
                // (nchunk + 2) % 4, those of waves 2, 3 to (nchunk + 3) % 4
                const unsigned dump0 = lds0 + ((nchunk + 2 + (wave >> 1)) & 3) * BUF_BYTES + (wave & 1) * DUMP_WAVE_BYTES;
#pragma unroll
                for (int i = 0; i < NTS; ++i) *(lds_fw)(size_t)(dump0 + i * 256 + lane * 4) = X[i];
                lds_wait();
                CELLS_STAMP(5)  // barrier + dump

                // ---- plane loop (straight line; the LDS operands of plane j + 1 are in flight during plane j) -----
                {
                    const unsigned gA = lds0 + (nchunk & 3) * BUF_BYTES, gB = lds0 + ((nchunk + 1) & 3) * BUF_BYTES;
                    const unsigned xq = dump0 + q * 16;
                    struct PlaneOps { float X0, X1, X2, X3; v4f G00, G01, G10; float N11, D2; };
                    auto fetch = [&](int info) {
                        const int texi = (info >> 20) & 1023;
                        const int tex = texi == 1023 ? 0 : texi;
                        PlaneOps o;
                        o.X0 = *(lds_f)(size_t)(xq + ((info & 31) << 8));
                        o.X1 = *(lds_f)(size_t)(xq + (((info >> 5) & 31) << 8) + 4);
                        o.X2 = *(lds_f)(size_t)(xq + (((info >> 10) & 31) << 8) + 8);
                        o.X3 = *(lds_f)(size_t)(xq + (((info >> 15) & 31) << 8) + 12);
                        o.G00 = *(lds_v4f)(size_t)(gA + tex * 16);
                        o.G01 = *(lds_v4f)(size_t)(gA + tex * 16 + 16);
                        o.G10 = *(lds_v4f)(size_t)(gA + (tex + pitch) * 16);
                        o.N11 = *(lds_f)(size_t)(gA + (tex + pitch) * 16 + 16);
                        o.D2 = *(lds_f)(size_t)(gB + tex * 4);
                        return o;
                    };
#if FCELLS_PREFETCH
                    PlaneOps cur = fetch(ki[0]), nxt = cur;
#endif
#pragma unroll
                    for (int j = 0; j < JS; ++j) {
#if FCELLS_PREFETCH
                        if (j + 1 < JS) nxt = fetch(ki[j + 1 < JS ? j + 1 : j]);
#else
                        const PlaneOps cur = fetch(ki[j]);
#endif
                        // (a memory clobber, not only a scheduling fence: instruction selection otherwise hoists the LDS
                        //  reads of all 16 planes to the top of this straight-line block and spills them)
                        asm volatile("" ::: "memory");
                        __builtin_amdgcn_sched_barrier(0);
                        const int info = ki[j];
                        const bool any = ((info >> 20) & 1023) != 1023;
                        float fw = fx[j], fe = 1.0f - fw, fn = fy[j], fs = 1.0f - fn;
                        if (!any) { fw = fw * 0.0f; fe = fe * 0.0f; fn = fn * 0.0f; fs = fs * 0.0f; }  // NaN stays NaN, like ATen
                        // |sum_t w_t s_t|^2, separable in the x weights (e, w) and the y weights (s, n)
                        const float ee = fe * fe, ww = fw * fw, ew = fe * fw;
                        const float A = ee * cur.G00.x + ww * cur.G01.x + 2.0f * ew * cur.G00.y;       // top row:    N00, N01, H00
                        const float B = ee * cur.G10.x + ww * cur.N11 + 2.0f * ew * cur.G10.y;         // bottom row: N10, N11, H10
                        const float Cq = ee * cur.G00.z + ww * cur.G01.z + ew * (cur.G00.w + cur.D2);  // cross rows: V00, V01, D1 + D2
                        const float Q = (fs * fs) * A + (fn * fn) * B + 2.0f * (fs * fn) * Cq;
                        // X by role: the even column is the left one iff x0 is even (bit 30), the even row the top one
                        // iff y0 is even (bit 31)
                        const bool xodd = (info & (1 << 30)) != 0, yodd = info < 0;
                        const float wxe = xodd ? fw : fe, wxo = xodd ? fe : fw;
                        const float wye = yodd ? fn : fs, wyo = yodd ? fs : fn;
                        const float XW = (wye * wxe) * cur.X0 + (wye * wxo) * cur.X1 + (wyo * wxe) * cur.X2 + (wyo * wxo) * cur.X3;
                        const float c = div_sigma((Q - 2.0f * XW) + rr);
                        float& o = cost[m * JS + j];
                        o = (!MV || v == 0) ? (0.0f + c) : (o + c);
                        asm volatile("" : "+v"(o));  // (or the whole evaluation is sunk into the epilogue, operands spilled)
#if FCELLS_PREFETCH
                        cur = nxt;
#endif
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                CELLS_STAMP(6)  // plane loop
            }  // windows
        }      // views

        if (bail) {
            if (tid == 0) {  // flag the tile and append it to the generic kernel's work list (count: queue[8])
                tile_flags[b * ntile + tile] = 1;
                redo_list[atomicAdd(&queue[8], 1)] = b * ntile + tile;
            }
        } else {
            // ---- epilogue from registers: cost store, log-softmax over D, expectation ------------------------
            const size_t obase = (size_t)b * a.D * HW + p;
            float* const cost_out = CELLS_ARG(float*, cost_out);
            float* const logp_out = CELLS_ARG(float*, logp_out);
            float* const depth_out = CELLS_ARG(float*, depth_out);
            const int ke = opaque_v(JS * r);
            if (cost_out && live) {
#pragma unroll
                for (int i = 0; i < NWIN * JS; ++i) cost_out[obase + (size_t)((i / JS) * 4 * JS + ke + (i % JS)) * HW] = cost[i];
            }
            if (logp_out || depth_out) {
                float mx = -INFINITY;
#pragma unroll
                for (int i = 0; i < NWIN * JS; ++i) mx = fmaxf(mx, cost[i]);
                mx = fmaxf(mx, CELLS_DPP_F(mx, QP_XOR1));
                mx = fmaxf(mx, CELLS_DPP_F(mx, QP_XOR2));
                // p_k = e_k / s with e_k = exp(c_k - max): one exp per plane; log p_k = (c_k - max) - log s
                float ssum = 0.0f, esum = 0.0f;
#pragma unroll
                for (int m = 0; m < NWIN; ++m) {
                    v4f d4[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) d4[i] = *(lds_v4f)(size_t)(lds0 + DTAB_OFF + (m * 4 * JS + ke + 4 * i) * 4);
#pragma unroll
                    for (int j = 0; j < JS; ++j) {
                        const float ek = exp_nonpos(cost[m * JS + j] - mx);
                        ssum = ssum + ek;
                        esum = __builtin_fmaf(d4[j >> 2][j & 3], ek, esum);
                        if ((j & 3) == 3) __builtin_amdgcn_sched_barrier(0);
                    }
                }
                ssum = ssum + CELLS_DPP_F(ssum, QP_XOR1);
                ssum = ssum + CELLS_DPP_F(ssum, QP_XOR2);
                esum = esum + CELLS_DPP_F(esum, QP_XOR1);
                esum = esum + CELLS_DPP_F(esum, QP_XOR2);
                const float ls = logf(ssum);
                if (logp_out && live) {
#pragma unroll
                    for (int i = 0; i < NWIN * JS; ++i)
                        logp_out[obase + (size_t)((i / JS) * 4 * JS + ke + (i % JS)) * HW] = (cost[i] - mx) - ls;
                }
                if (depth_out && live && r == 0) depth_out[(size_t)b * HW + p] = esum / ssum;
            }
        }
        CELLS_STAMP(7)  // epilogue
        __syncthreads();
        item_par ^= 1;
        item = s_item[item_par];
        CELLS_STAMP(8)
    }
#ifdef CELLS_STAMPS
    if (lane == 0)
        for (int i = 0; i < 13; ++i) atomicAdd(reinterpret_cast<unsigned long long*>(queue + 16) + i, stamp_acc[i]);
#endif
}

size_t sweep_cells_fast_lds_bytes() { return (size_t)fast::DTAB_OFF + 512; }

// Launches the straight-line kernel on every tile (persistent grid of 2 blocks per CU).  D must be 64 or 128.
hipError_t launch_sweep_cells_fast(const SweepArgs& a, const float4* packed, int* flags, int* queue, int* redo_list,
                                   int tiles_x, int tiles, int n_cu, hipStream_t stream) {
    using namespace fast;
    int nblk = (n_cu * 2 + 7) & ~7;
    const long long full = 8ll * ((tiles + 7) / 8) * a.B;
    if (full <= nblk) nblk = (int)full;
    typedef void (*kern_t)(SweepArgs, const float4*, int*, int*, int*, int, int);
    static const kern_t kerns[4] = {sweep_cells_fast_kernel<1, false>, sweep_cells_fast_kernel<1, true>,
                                    sweep_cells_fast_kernel<2, false>, sweep_cells_fast_kernel<2, true>};
    const int which = (a.D <= 64 ? 0 : 2) + (a.V > 1 ? 1 : 0);
    const size_t lds = sweep_cells_fast_lds_bytes();
    // (the attribute is sticky per kernel and device-independent in value: setting it on every launch costs a few
    //  hundred ns and keeps this function free of per-device state)
    hipError_t e = hipFuncSetAttribute((const void*)kerns[which], hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kerns[which], dim3(nblk), dim3(NT), lds, stream, a, packed, flags, queue, redo_list, tiles_x, tiles);
    return hipGetLastError();
}

}  // namespace pdepth
