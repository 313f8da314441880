// LDS-tiled plane sweep: the fast path of pdepth_sweep_{cost,dpv}_f32.
//
// Same bit-faithful sample positions and bilinear weights as sweep_direct.hip, but the source comes from LDS
// instead of global memory, and the planes of a tile are evaluated in one of two forms:
//
//   DIRECT groups (near planes, which move the sample by texels per plane): per (pixel, plane, channel) four taps
//   and five fma-class ops, diff = fma(s00,nw,-r) ... fma(s11,se,.), acc = fma(diff,diff,acc);
//
//   the BAND group (all planes from ks on, which move the sample by a fraction of a texel per plane; L2 only):
//   the taps of all those planes of one pixel fall into a box of at most 64 source texels, and
//       sum_c (sum_t w_t s_t[c] - r[c])^2 = w^T G w - 2 sum_t w_t X_t + |r|^2
//   needs the channel loop only for X_t = <r, s_t>, ONE dot product per (pixel, box texel) instead of one
//   interpolation per (pixel, plane); G (Gram terms of neighbouring source texels) comes from the pre-pass.
//   On the benchmark poses 40..64 of the 64 planes are band planes and a pixel's box has ~30 texels (against
//   160..256 taps): ~2x fewer VALU and LDS instructions per tile than direct evaluation of everything.
//
//   grid    = persistent: as many blocks as the chip holds at once (3 per CU), each pulling (batch item, tile)
//             work items from a per-XCD queue counter in the workspace, so the per-block setup (depth candidates
//             to LDS, suffix min/max) and the launch cost are paid once per block instead of once per tile (-3.5 %);
//             when the grid covers all items anyway (small problems) block i just takes item i, no atomics;
//   block   = NSUB tiles of 16x4 reference pixels, side by side, x 4 waves each; the 4 waves of a tile hold the same
//             64 pixels (lane = pixel).  NSUB = 1: 256 threads, 3 blocks per CU.  NSUB = 2: 512 threads, 2 blocks
//             per CU (16 instead of 12 waves); the two tiles share band decision, window search, staged windows
//             and barriers, and keep their own geometry, X, costs and epilogue.
//             Direct group: wave w owns KP = 4 (or 2, 1) consecutive planes of a group of 16 (8, 4); their
//             geometry (LDS tap address + 4 weights) lives in registers, computed once per (pixel, plane, view).
//             (8 planes per wave and groups of 32 were measured: +28 VGPRs, and slower on every configuration
//             once the band mode had taken the far planes.)
//             Band group: wave w accumulates box texels w, w+4, ... (X in registers), the waves exchange X
//             through LDS, then wave w combines planes ks+w, ks+w+4, ... (geometry recomputed on the fly);
//   source  = a pre-pass (pack_c4_kernel) re-lays every source view from NCHW to channel-group-planar
//             [C/4 + 2][H][W] float4 texels in the workspace: C/4 planes of 4 channels (zero padded), then the
//             two Gram planes;
//   window  = bounding box of every texel the block touches in the current group, staged four channels at a
//             time into LDS as float4 texels [row][col] by LDS-DMA (buffer_load_dwordx4 ... lds: a
//             wave-instruction moves 64 texels = 1 KiB, no registers, no ds_write).  Direct groups: two 16 KB
//             buffers, chunk ch+1 in flight while chunk ch is computed, one barrier per chunk.  Band group
//             (small windows): the same 32 KB as a ring of 3 stages of two chunks each, two stages in flight.
//             The DMA is issued from inline asm with hand-counted s_waitcnt, because the compiler would
//             otherwise drain it in front of every ds_read.  Texels outside the image are fetched with an
//             out-of-range buffer offset and arrive as zeros, which IS padding_mode='zeros';
//   taps    = (direct) software pipelined over the wave's planes: the four ds_read_b128 of plane i+1 are in
//             flight while the 20 fma of plane i execute; the window buffer is an immediate of the ds_read;
//   ref     = the tile's reference features of the chunk (wave w moves channel 4*ch+w by LDS-DMA from NCHW);
//   costs   = cost[k][pixel] of the tile in LDS (D x 64 floats), accumulated over views in view order like
//             homography.py:129; the fused epilogue (log_softmax over D + E[d]) runs on those with the 4 waves
//             splitting the planes, so nothing but the requested outputs is ever written to HBM.
//
// A block whose direct window does not fit (extreme poses) first splits the group into 2 or 4 parts of
// consecutive planes, then raises its tile flag and leaves the tile to the gather kernel of sweep_direct.hip; so
// does a band group whose box assumption is violated (never observed).  The three evaluations agree to rounding:
// ~3e-7 (direct) and ~5e-7 (band) of the largest cost of the volume, i.e. an ulp or two of the fp32 cost.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "geometry.hpp"
#include "kernels.hpp"
#include "pick.hpp"

namespace pdepth {

namespace {

#ifndef PDEPTH_NSUB
#define PDEPTH_NSUB 1
#endif
constexpr int TW = 16, TH = 4;    // sub-tile (pixels); one wave covers it
constexpr int NSUB = PDEPTH_NSUB; // sub-tiles per block, side by side: they share decision, windows, staging, barriers
constexpr int NPG = 4;            // plane groups (= waves) per sub-tile
constexpr int NW = NPG * NSUB;    // waves per block
constexpr int NT = 64 * NW;       // threads per block
constexpr int KP = 4;             // planes per group
constexpr int SG = NPG * KP;      // planes per super group
constexpr int NBUF = 2;           // LDS window buffers: chunk ch+1 is in flight (LDS-DMA) while chunk ch is computed
#define PDEPTH_NTEX (PDEPTH_NSUB == 1 ? 1024 : 1216)   // (build constants of the two variants; their tuning history: DESIGN.md of rounds 1-3)
#define PDEPTH_OCC (PDEPTH_NSUB == 1 ? 3 : 4)
constexpr int NTEX_MAX = PDEPTH_NTEX;    // window texels per buffer (LDS: NTEX_MAX*NBUF*16 B = 32 KB)
constexpr int BUF_BYTES = NTEX_MAX * 16;   // one window buffer of the direct groups
// Band mode (see "band group" in the kernel).  Its windows are small, so the 32 KB of the two direct-mode window
// buffers (plus nothing else) are re-cut into a ring of BR stages of two channel groups each:
//     [BR stages][2 chunks][BAND_TEX texels * 16 B]   window ring   (24 KB)
//     [2 * BR chunks][4 channels][64 pixels] floats   reference ring (6 KB)
// and, once the channel loop is done, the last stage -- the two Gram planes -- sits in ring slot 0 and everything
// behind it (the other slots and the reference ring) holds the X exchange buffer (NX_MAX slots of 256 B).
#define PDEPTH_BR (PDEPTH_NSUB == 1 ? 3 : 2)
#define PDEPTH_BAND_TEX (PDEPTH_NSUB == 1 ? 256 : 448)
// Band boxes follow the epipolar line row by row (1) or are bounding rectangles (0).  Only where the X slots are scarce:
// the two-tile build has 48 per pixel, and the rectangle of planes [16, D) exceeds them in 17 % of the tiles of the
// benchmark pose (=> twice the direct work); with 64 slots (one-tile build) the rectangle fits and the shear only costs.
#define PDEPTH_SHEAR (PDEPTH_NSUB == 2)
#define PDEPTH_NX (PDEPTH_NSUB == 1 ? 64 : 48)
constexpr int BR = PDEPTH_BR;               // ring depth in stages (BR-1 stages in flight)
constexpr int BAND_TEX = PDEPTH_BAND_TEX;    // window texels of a band group
constexpr int BAND_CHUNK_BYTES = BAND_TEX * 16;
constexpr int BAND_STAGE_BYTES = 2 * BAND_CHUNK_BYTES;
constexpr int BAND_REF_OFF = BR * BAND_STAGE_BYTES;   // 24576 (one DMA instruction per wave and chunk covers 256 texels)
constexpr int BAND_REF_CHUNK = 1024 * NSUB;           // reference features of one chunk: [NSUB][4 channels][64 pixels]
constexpr int NX_MAX = PDEPTH_NX;            // box texels per pixel: X exchange buffer = NX_MAX * 256 B
constexpr int XPW = (NX_MAX + NPG - 1) / NPG;  // X accumulators per wave
static_assert(BAND_REF_OFF + 2 * BR * BAND_REF_CHUNK <= 2 * BUF_BYTES, "band ring must fit the direct-mode window buffers");
// after the channel loop: [Gram stage = ring slot 0][X exchange buffer over the other slots and the reference ring]
static_assert(BAND_STAGE_BYTES + NSUB * NX_MAX * 256 <= 2 * BUF_BYTES, "X exchange buffer does not fit");
static_assert(BAND_TEX <= NT, "one window DMA instruction per wave and chunk");
constexpr int SLOTS = (NTEX_MAX + NT - 1) / NT;          // sub-blocks of a window (256 texels each = one DMA pass of the block)

// Wave-wide min / max with a scalar result, for fully active waves: four DPP steps make every row of 16 lanes
// uniform (xor 1, xor 2, mirror within 8, mirror within 16), the four rows are combined on the scalar unit.
// (__shfl_xor reductions go through the LDS crossbar, 6 dependent ds_bpermute each.)
#define PDEPTH_DPP_STEP(OP, ctrl) v = OP(v, __builtin_amdgcn_update_dpp(v, v, ctrl, 0xf, 0xf, false))
__device__ __forceinline__ int wave_min_s(int v) {
    PDEPTH_DPP_STEP(min, 0xB1); PDEPTH_DPP_STEP(min, 0x4E); PDEPTH_DPP_STEP(min, 0x141); PDEPTH_DPP_STEP(min, 0x140);
    return min(min(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)),
               min(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}
__device__ __forceinline__ int wave_max_s(int v) {
    PDEPTH_DPP_STEP(max, 0xB1); PDEPTH_DPP_STEP(max, 0x4E); PDEPTH_DPP_STEP(max, 0x141); PDEPTH_DPP_STEP(max, 0x140);
    return max(max(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)),
               max(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}
#undef PDEPTH_DPP_STEP

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
// s_waitcnt vmcnt(n) for a wave-uniform run-time n in {0, 2, 4, 8} (the immediate must be a constant)
__device__ __forceinline__ void wait_dma_but(int n) {
    if (n == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (n == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if (n == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
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

// A kernel argument that is used once or twice per tile and would otherwise occupy scalar registers for the whole
// kernel (which already spills SGPRs into VGPR lanes): re-read it from the kernarg segment at the point of use.
// `a` is the first kernel argument, so field offsets are offsets into the segment.
template <typename T>
__device__ __forceinline__ T cold_arg(size_t offset) {
    typedef const char __attribute__((address_space(4))) * kptr;
    typedef const volatile T __attribute__((address_space(4))) * vptr;
    return *(vptr)((kptr)__builtin_amdgcn_kernarg_segment_ptr() + offset);
}
#define PDEPTH_COLD_ARG(type, field) cold_arg<type>(offsetof(SweepArgs, field))

// The file is compiled twice (Makefile): PDEPTH_NSUB = 1 (one 16x4 tile per block; also defines the shared
// pre-pass and workspace helpers) and PDEPTH_NSUB = 2 (two tiles side by side per block).  Each variant keeps its
// kernel in its own namespace; launch_sweep_tiled() (first variant) picks one per call.
#define PDEPTH_CAT_(a, b) a##b
#define PDEPTH_CAT(a, b) PDEPTH_CAT_(a, b)
#define PDEPTH_VARIANT PDEPTH_CAT(tiled_n, PDEPTH_NSUB)
namespace PDEPTH_VARIANT {

// SPEC: the evaluation configuration -- D = 64 planes, C = 67 channels, one source view -- as compile-time constants
// (loop bounds, the chunk count, the LDS carve-up and the view loop fold; the shape of the image stays a run-time value).
template <int METRIC, bool SPEC>
__global__ __launch_bounds__(NT, PDEPTH_OCC) void sweep_tiled_kernel(SweepArgs a, const float4* __restrict__ packed,
                                                              int* __restrict__ tile_flags, int* __restrict__ queue,
                                                              int tiles_x, int ntile, const float* __restrict__ route_stats) {
    if (PDEPTH_COLD_ARG(int, pick) == PICK_SKIP_IF_SET && queue[PICK_SLOT] != 0) return;   // (the pre-pass chose the other kernel: pick.hpp)
    if (poison_on_foreign_layout(a, queue, LAYOUT_C4)) return;
    const int aD = SPEC ? 64 : a.D, aC = SPEC ? 67 : a.C, aV = SPEC ? 1 : a.V;
    extern __shared__ __attribute__((aligned(16))) float4 lds4[];
    float4* win = lds4;                                   // [NBUF][NTEX_MAX]
    float* reft = reinterpret_cast<float*>(lds4 + NBUF * NTEX_MAX);  // [NBUF][NSUB][4 channels][64 pixels]
    float* costs_all = reft + NBUF * NSUB * 256;          // [NSUB][D][64]
    float* red_all = costs_all + (size_t)NSUB * aD * 64; // [NSUB][NPG][64]
    float* dcl = red_all + NW * 64;                       // [D] depth candidates (read wave-uniformly, per plane)
    float* dlo = dcl + aD;                               // [D/8 + 1] min of d_candi[8 j .. D)
    float* dhi = dlo + (aD / 8 + 1);                     // [D/8 + 1] max of d_candi[8 j .. D)
    __shared__ int s_bbox[2][NW][4];   // per-wave bounding boxes of window_of(), double buffered by call parity
    __shared__ int s_dec[2][NW][8];    // per-wave band-decision values (NSUB > 1: the sub-tiles see different pixels)
    __shared__ int s_item[2];          // work item of this block: current / prefetched next
    int bbox_parity = 0, dec_parity = 0;

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform
    const int sub = wave / NPG;      // sub-tile of this wave
    const int pgl = wave % NPG;      // plane group of this wave within its sub-tile
    float* costs = costs_all + (size_t)sub * aD * 64;    // [D][64] of this sub-tile
    float* red = red_all + sub * NPG * 64;                // [NPG][64] of this sub-tile
    const int lane = tid & 63;       // pixel of the tile
    const int lx = lane & 15, ly = lane >> 4;
    // Persistent blocks: the grid is sized to fill the chip once and every block pulls (tile, batch item) work
    // items from a queue, so the per-block setup below is paid once per block and not once per tile.  XCD-aware:
    // workgroups are dealt round-robin over the 8 XCDs, so blocks i and i+8 share an L2.  Every XCD owns one
    // contiguous band of tiles (its own queue counter): neighbouring tiles stage overlapping source windows,
    // which then hit that XCD's L2 instead of going out to the Infinity Cache.
    // A block whose own band is exhausted takes items of the other XCDs' bands (round robin from its own): the bands
    // are strips of the image and do not cost the same -- in a forward motion the strips at the top and bottom see the
    // largest disparities -- and the last item of a block would otherwise leave its CU idle for up to an item's time.
    const int xcd = blockIdx.x & 7, qq = ntile >> 3, rr = ntile & 7;
    auto band_first_of = [&](int x) { return x < rr ? x * (qq + 1) : rr * (qq + 1) + (x - rr) * qq; };
    auto band_tiles_of = [&](int x) { return qq + (x < rr ? 1 : 0); };
    const bool colmajor = rr == 0 && qq % tiles_x == 0;  // the band is a whole number of tile rows
    const int HW = a.H * a.W;
    const int nchunk = (aC + 3) / 4;
    const float half_w = (float)a.W / 2.0f, half_h = (float)a.H / 2.0f;
    // cost / sigma with the divide chain of geometry.hpp (bit-identical to the IEEE divide for finite operands in
    // range); non-finite costs take the real divide so that inf / NaN come out exactly as before
    const float sigma = a.sigma, rsigma = refined_rcp(a.sigma);
    auto div_sigma = [&](float v) {
        float q = div_core(v, sigma, rsigma);
        if (!(fabsf(v) < 1.0e30f)) q = v / sigma;
        return q;
    };
    for (int k = tid; k < aD; k += NT) dcl[k] = a.d_candi[k];
    __syncthreads();
    {   // suffix min / max of the depth candidates per 8 planes: one wave, 8 lanes per segment
        const int nseg = aD / 8 + 1;
        if (tid < 64) {
            for (int s0 = 0; s0 < nseg; s0 += 8) {
                const int seg = s0 + (tid >> 3), k = seg * 8 + (tid & 7);
                float lo = (seg < nseg && k < aD) ? dcl[k] : INFINITY, hi = (seg < nseg && k < aD) ? dcl[k] : -INFINITY;
#pragma unroll
                for (int sh = 4; sh >= 1; sh >>= 1) { lo = fminf(lo, __shfl_xor(lo, sh)); hi = fmaxf(hi, __shfl_xor(hi, sh)); }
                if ((tid & 7) == 0 && seg < nseg) { red_all[seg] = lo; red_all[64 + seg] = hi; }  // (free scratch here)
            }
        }
        __syncthreads();
        if (tid < nseg) {
            float lo = INFINITY, hi = -INFINITY;
            for (int j = tid; j < nseg; ++j) { lo = fminf(lo, red_all[j]); hi = fmaxf(hi, red_all[64 + j]); }
            dlo[tid] = lo; dhi[tid] = hi;
        }
        __syncthreads();
    }
    // (the pre-pass found channel offsets larger than the spread of the features: the correlation form of the band group would
    //  cancel -- every plane directly, in the reference's form; sweep_pack.hip)
    const bool noband = queue[NONCENTRED_SLOT] != 0;
    const int win_lds0 = (int)lds_addr_of(win);
    // work item -> batch item, tile, this lane's pixel
    auto map_item = [&](int item_, int& b_, int& tile_, bool& live_, int& p_) {
        const int qx = item_ >> 24, it = item_ & 0xffffff;   // queue (XCD band) the item came from, index in it
        const int band_tiles = band_tiles_of(qx);
        b_ = it / band_tiles;
        const int ti = it - b_ * band_tiles;
        tile_ = band_first_of(qx) + ti;
        const int tiles_y_ = ntile / tiles_x;
        if (rr == 0 && tiles_y_ % 16 == 0 && tiles_y_ * tiles_x == ntile) {
            // XCD q owns half-bands q and 8 + q of the image's 16 (as the matrix-pipe kernel, sweep_mfma.hip): on a forward
            // motion the cost of a tile grows with its distance from the image centre, and this way every XCD gets the same
            // mix; the heavier half first and, inside a half, columns from both image borders inwards.
            const int hb_rows = tiles_y_ / 16, half_tiles = hb_rows * tiles_x;
            const int second = ti >= half_tiles ? 1 : 0, tih = ti - second * half_tiles;
            const int hbi = (qx < 4) == (second == 0) ? qx : 8 + qx;
            const int cc = tih / hb_rows, r_ = tih - cc * hb_rows;
            const int col = (cc & 1) ? tiles_x - 1 - (cc >> 1) : (cc >> 1);
            tile_ = (hbi * hb_rows + r_) * tiles_x + col;
        } else
        if (colmajor) {
            // walk the band column by column, so that the blocks in flight on one XCD share a narrow strip of
            // source columns (working set ~1 MB instead of the full image width)
            const int band_rows = qq / tiles_x;
            tile_ = (qx * band_rows + ti % band_rows) * tiles_x + ti / band_rows;
        }
        const int x = (tile_ % tiles_x) * (TW * NSUB) + sub * TW + lx, y = (tile_ / tiles_x) * TH + ly;
        live_ = x < a.W && y < a.H;
        p_ = live_ ? y * a.W + x : (min(y, a.H - 1) * a.W + min(x, a.W - 1));
    };
    // (when the grid already covers every item -- small problems -- block i simply takes item i of its XCD:
    //  no atomics on the critical path of a launch that is latency bound anyway)
    const bool queued = (int)gridDim.x < 8 * ((ntile + 7) / 8) * a.B;  // else: one block per item of every XCD band
    // Next item of this block.  The atomic on the block's own queue is ISSUED at the top of a tile and its result is only
    // looked at at the end of the tile (the latency rides under the tile's first DMA wait instead of holding wave 0,
    // and with it the block's first barrier, for a round trip to L2).  Only when the own band is exhausted -- at the
    // end of a launch -- the other bands are polled, and only those that have more items left than their own XCD
    // runs blocks: the last round of a band is quicker on its own XCD (warm L2) than spread over the others.
    const int n_own = band_tiles_of(xcd) * a.B;
    auto steal = [&]() -> int {   // (one thread) index | queue << 24, or -1 when every band is exhausted
#pragma unroll 1
        for (int j = 1; j < 8; ++j) {
            const int x = (xcd + j) & 7, n = band_tiles_of(x) * a.B;
            if (n - *(volatile int*)&queue[x] <= 2 * (int)(gridDim.x >> 3)) continue;
            const int it = atomicAdd(&queue[x], 1);
            if (it < n) return it | (x << 24);
        }
        return -1;
    };
    int nxt_own = n_own;   // (thread 0) result of the atomic issued at the top of the tile
    bool own_done = false; // (thread 0) the own band is exhausted
    if (queued) {
        if (tid == 0) {
            const int it = atomicAdd(&queue[xcd], 1);
            own_done = it >= n_own;
            s_item[0] = own_done ? steal() : (it | (xcd << 24));
        }
        __syncthreads();
    }
    int item = __builtin_amdgcn_readfirstlane(queued ? s_item[0] : ((int)(blockIdx.x >> 3) < n_own ? (int)(blockIdx.x >> 3) | (xcd << 24) : -1));
    int item_par = 0;
    while (item >= 0) {
    if (tid == 0 && queued && !own_done) nxt_own = atomicAdd(&queue[xcd], 1);
    int b, tile, p; bool live;
    map_item(item, b, tile, live, p);
    // the gather kernel's flags are per 16x4 tile: this wave's sub-tile (if it lies in the image at all)
    auto flag_subtile = [&]() {
        const int tiles16_x = (a.W + TW - 1) / TW, tiles_y = (a.H + TH - 1) / TH;
        const int tx16 = (tile % tiles_x) * NSUB + sub, ty = tile / tiles_x;
        if (tx16 < tiles16_x) {
            tile_flags[b * tiles16_x * tiles_y + ty * tiles16_x + tx16] = 1;
            atomicAdd(&queue[GATHER_COUNT_SLOT], 1);
        }
    };

    const float* const cxcy_ = PDEPTH_COLD_ARG(const float*, cxcy);   // (per-tile arguments are re-read from the kernarg
    const float* const rays_ = PDEPTH_COLD_ARG(const float*, rays);   //  segment instead of living in SGPRs across the tile)
    const float cx = cxcy_[b * 2 + 0], cy = cxcy_[b * 2 + 1];
    const float rcx = refined_rcp(cx), rcy = refined_rcp(cy);
    const float r0 = rays_[((size_t)b * 3 + 0) * HW + p];
    const float r1 = rays_[((size_t)b * 3 + 1) * HW + p];
    const float r2 = rays_[((size_t)b * 3 + 2) * HW + p];
    const float* refb = PDEPTH_COLD_ARG(const float*, ref) + (size_t)b * PDEPTH_COLD_ARG(long long, ref_bstride);
    const v4i ref_rsrc = make_rsrc(refb, aC * HW * 4);

    // Routing (route_stats = the statistics rows; flag 1 set by the pre-pass / flag clear of the call, sweep_pack.hip): an ill-conditioned item
    // is the gather kernel's, whole -- every tile of it is handed over as a tile whose windows do not fit is
    const bool routed = route_stats && reinterpret_cast<const int*>(route_stats + (size_t)b * STATS_STRIDE + STATS_FLAGS)[1] != 0;
    for (int v = 0; v < aV; ++v) {
        if (routed) {   // block-uniform
            if (lane == 0 && pgl == 0) flag_subtile();
            goto tile_done;
        }
        ViewXform xf;
        make_view_xform(PDEPTH_COLD_ARG(const float*, K) + b * 9, PDEPTH_COLD_ARG(const float*, R) + ((size_t)b * aV + v) * 9,
                        PDEPTH_COLD_ARG(const float*, t) + ((size_t)b * aV + v) * 3, PDEPTH_COLD_ARG(int, blas_mode), xf);
        float t2a, t2b, t2c;
        ray_term2(xf, r0, r1, r2, t2a, t2b, t2c);
        // [C/4 + 2][H][W] float4 texels: the channel groups, then the two Gram planes of the band mode
        const float4* srcv = packed + ((size_t)b * aV + v) * (nchunk + 2) * HW;
        const v4i src_rsrc = make_rsrc(srcv, (nchunk + 2) * HW * 16);

        // ---- band decision -------------------------------------------------------------------------
        // Far planes move the sample by a fraction of a texel per plane: the taps of planes [ks, D) of one
        // pixel all fall into a small box of source texels (a few columns x rows), far fewer texels than
        // 4 taps x planes.  For those planes the L2 cost is evaluated in its correlation form
        //     sum_c (sum_t w_t s_t[c] - r[c])^2 = w^T G w - 2 sum_t w_t X_t + |r|^2,
        // X_t = <r, s_t> accumulated ONCE per (pixel, box texel) over the channels, G = Gram terms of the
        // source texels (precomputed by the pre-pass, two extra planes of the packed source).  The box of a
        // pixel is spanned by the positions at the smallest and largest depth of the range (the position
        // moves monotonically along the epipolar line between them unless the plane through the camera
        // centre is crossed -- checked), and every footprint is verified against it in the combine step; a
        // violation hands the tile to the gather kernel.  ks = first multiple of 16 whose box has at most
        // NX_MAX texels for every pixel of the tile; planes [0, ks) are evaluated directly.  L1 has no such form.
        int ks = aD;             // block-uniform: first plane of the band group (D: none)
        int bbx0 = 0, bby0 = 0;   // per pixel: top-left texel of its box
        int NC = 0, NR = 0;       // block-uniform box size
        int bsh0 = 0, bsh1 = 0;   // per pixel: column offset of box row r (4 bits each, rows 0..7 / 8..15): the box is SHEARED
        int gwx0 = 0, gwy0 = 0, gWC = 0, gWR = 0;  // staged window of the band group
        if (METRIC == 0 && !noband) {
            int t_x0 = 0, t_y0 = 0, t_nc = 0, t_nr = 0, t_wx0 = 0, t_wy0 = 0, t_wc = 0, t_wr = 0, t_sh0 = 0, t_sh1 = 0;
            // one candidate: does the band group [kc, D) fit?  (block-uniform result; contains a barrier for NSUB > 1)
            auto trial = [&](int kc) -> bool {
                const float dl = dlo[kc >> 3], dh = dhi[kc >> 3];
                float ixl, iyl, ixh, iyh;
                plane_sample_pos_fast(xf, t2a, t2b, t2c, dl, cx, cy, rcx, rcy, half_w, half_h, ixl, iyl);
                plane_sample_pos_fast(xf, t2a, t2b, t2c, dh, cx, cy, rcx, rcy, half_w, half_h, ixh, iyh);
                const float denl = (xf.kt[2] + t2c * dl) + 1e-10f, denh = (xf.kt[2] + t2c * dh) + 1e-10f;
                const float big = fmaxf(fmaxf(fabsf(ixl), fabsf(iyl)), fmaxf(fabsf(ixh), fabsf(iyh)));
                bool ok = denl * denh > 0.0f && big < 1.0e6f;  // false for NaN as well
                const int x0 = (int)floorf(fminf(ixl, ixh) - 1e-3f), x1 = (int)floorf(fmaxf(ixl, ixh) + 1e-3f) + 1;
                const int y0 = (int)floorf(fminf(iyl, iyh) - 1e-3f), y1 = (int)floorf(fmaxf(iyl, iyh) + 1e-3f) + 1;
#if PDEPTH_SHEAR
                // Sheared box: the samples lie on the straight segment between the two end positions, so texel row
                // y0 + r is touched only by the part of the segment with iy in [y0 + r - 1, y0 + r + 1]; its
                // columns start at x0 + shift[r] and the box is NR rows of NC columns, NC = the widest row of any
                // pixel -- for a diagonal epipolar line far fewer texels than the bounding rectangle.
                int shl = 0, shh = 0, ncs = 0, xsm = x0;
                {
                    const float dxs = ixh - ixl, dys = iyh - iyl;
                    const bool horiz = !(fabsf(dys) > 1e-4f);
                    const float inv = horiz ? 0.0f : __builtin_amdgcn_rcpf(dys);
                    const int nrw = wave_max_s(ok ? y1 - y0 + 1 : 0);
                    bool shok = true;
                    const bool flat = __builtin_amdgcn_ballot_w64(ok && !horiz) == 0ull;   // every segment horizontal: plain rectangles
                    if (flat) { ncs = x1 - x0 + 1; }
                    for (int r = 0; r < 16 && r < nrw && !flat; ++r) {
                        const float rm = (float)(y0 + r - 1);
                        const float ta = (rm - iyl) * inv, tb = ((rm + 2.0f) - iyl) * inv;
                        const float lo = horiz ? 0.0f : fminf(fmaxf(fminf(ta, tb), 0.0f), 1.0f);
                        const float hi = horiz ? 1.0f : fminf(fmaxf(fmaxf(ta, tb), 0.0f), 1.0f);
                        const float xa = __builtin_fmaf(lo, dxs, ixl), xb = __builtin_fmaf(hi, dxs, ixl);
                        const int cmin = max((int)floorf(fminf(xa, xb) - 1e-3f), x0);
                        const int cmax = min((int)floorf(fmaxf(xa, xb) + 1e-3f) + 1, x1);
                        const bool rowv = ok && y0 + r <= y1;
                        const int sh = rowv ? cmin - x0 : 0;
                        ncs = max(ncs, rowv ? cmax - cmin + 1 : 0);
                        shok = shok && sh <= 15;
                        if (r < 8) shl |= (sh & 15) << (4 * r); else shh |= (sh & 15) << (4 * (r - 8));
                        xsm = max(xsm, rowv ? cmin : x0);
                    }
                    ok = ok && shok && nrw <= 16;
                }
                const int bw = ncs, bxs = xsm;    // width of the pixel's box, its rightmost row start
#else
                const int shl = 0, shh = 0;
                const int bw = x1 - x0 + 1, bxs = x0;
#endif
                int nc, nr, wx0_, wy0_, wc_, wr_;
                if (NSUB == 1) {  // every wave sees the same 64 pixels: no exchange needed
                    if (__builtin_amdgcn_ballot_w64(ok) != ~0ull) return false;  // some pixel crosses the pole / leaves the range
                    nc = wave_max_s(bw); nr = wave_max_s(y1 - y0 + 1);
                    if (nc > NX_MAX || nr > NX_MAX || nc * nr > NX_MAX) return false;   // (each factor first: the product of two huge boxes wraps)
                    wx0_ = wave_min_s(x0); wy0_ = wave_min_s(y0);
                    wc_ = wave_max_s(bxs) + nc - wx0_; wr_ = wave_max_s(y0) + nr - wy0_;
                } else {          // combine the sub-tiles through LDS (one barrier per trial, double buffered)
                    int (*sd)[8] = s_dec[dec_parity];
                    dec_parity ^= 1;
                    const int okw = __builtin_amdgcn_ballot_w64(ok) == ~0ull ? 1 : 0;
                    const int v1 = wave_max_s(ok ? bw : 0), v2 = wave_max_s(ok ? y1 - y0 + 1 : 0);
                    const int v3 = wave_min_s(ok ? x0 : 0), v4 = wave_min_s(ok ? y0 : 0);
                    const int v5 = wave_max_s(ok ? bxs : 0), v6 = wave_max_s(ok ? y0 : 0);
                    if (lane == 0) {
                        sd[wave][0] = okw; sd[wave][1] = v1; sd[wave][2] = v2; sd[wave][3] = v3; sd[wave][4] = v4;
                        sd[wave][5] = v5; sd[wave][6] = v6;
                    }
                    __syncthreads();
                    int allok = 1, xmax = INT_MIN, ymax = INT_MIN;
                    nc = 0; nr = 0; wx0_ = INT_MAX; wy0_ = INT_MAX;
#pragma unroll
                    for (int w2 = 0; w2 < NW; w2 += NPG) {  // one representative wave per sub-tile
                        allok &= sd[w2][0];
                        nc = max(nc, sd[w2][1]); nr = max(nr, sd[w2][2]);
                        wx0_ = min(wx0_, sd[w2][3]); wy0_ = min(wy0_, sd[w2][4]);
                        xmax = max(xmax, sd[w2][5]); ymax = max(ymax, sd[w2][6]);
                    }
                    if (!allok || nc > NX_MAX || nr > NX_MAX || nc * nr > NX_MAX) return false;
                    wc_ = xmax + nc - wx0_; wr_ = ymax + nr - wy0_;
                }
                if (wc_ > BAND_TEX || wr_ > BAND_TEX || wc_ * wr_ > BAND_TEX) return false;
                t_x0 = x0; t_y0 = y0; t_nc = nc; t_nr = nr; t_wx0 = wx0_; t_wy0 = wy0_; t_wc = wc_; t_wr = wr_;
                t_sh0 = shl; t_sh1 = shh;
                return true;
            };
            auto commit = [&](int kc) {
                ks = kc; bbx0 = t_x0; bby0 = t_y0; NC = t_nc; NR = t_nr; bsh0 = t_sh0; bsh1 = t_sh1;
                gwx0 = t_wx0; gwy0 = t_wy0; gWC = t_wc; gWR = t_wr;
            };
            // Candidates in the order of their likelihood: [16, D) first (the usual answer); [0, D) only if the box of
            // [16, D) leaves room (it contains it); later starts only if [16, D) does not fit.
            const int k1 = aD > 16 ? 16 : 0;
            if (trial(k1)) {
                commit(k1);
                // (with sheared boxes [0, D) only ever fits next to a box of [16, D) of at most a third of the slots:
                //  `tools/analysis`: 51 % of those tiles, 2 % of the tiles between a third and a half)
                if (k1 != 0 && NC * NR * (PDEPTH_SHEAR ? 3 : 2) <= NX_MAX && trial(0)) commit(0);
            } else {
                for (int kc = k1 + 16; kc < aD; kc += 16)
                    if (trial(kc)) { commit(kc); break; }
            }
        }
        const int kend = ks;  // planes [0, kend) are evaluated directly

        for (int k0 = 0; k0 < kend; k0 += SG) {
            // ---- geometry of this thread's KP planes (registers) ----------------------------
            // Plane assignment inside the 16-plane group: the KP = 4 planes of a wave are split into nsplit
            // parts of per = 4/nsplit planes; part q of wave w holds planes
            //     k0 + q*(4*per) + w*per + [0, per),
            // so the planes staged together (one part of all 4 waves) are 4*per CONSECUTIVE planes and
            // their common window shrinks with nsplit (nsplit = 1 is simply planes 4w .. 4w+3).
            int off[KP];  // packed (y0, x0) until the plane's part is staged, then its window texel index
            int kpl[KP];  // depth plane of slot i
            float wnw[KP], wne[KP], wsw[KP], wse[KP];
            auto geometry = [&](int nsplit_) {
                const int lper = nsplit_ == 1 ? 2 : nsplit_ == 2 ? 1 : 0;  // log2(planes per part and wave)
#pragma unroll
                for (int i = 0; i < KP; ++i) {
                    kpl[i] = k0 + ((i >> lper) << (lper + 2)) + (pgl << lper) + (i & ((1 << lper) - 1));
                    if (kpl[i] >= kend) {  // wave-uniform: not a plane of this group
                        off[i] = INT_MIN; wnw[i] = 0.0f; wne[i] = 0.0f; wsw[i] = 0.0f; wse[i] = 0.0f;
                        continue;
                    }
                    const int k = kpl[i];
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
                bx0 = wave_min_s(bx0); by0 = wave_min_s(by0); bx1 = wave_max_s(bx1); by1 = wave_max_s(by1);
                // one barrier per call: the slot written now was last read two calls ago, and every wave has
                // passed the barrier of the previous call since
                int (*sb)[4] = s_bbox[bbox_parity];
                bbox_parity ^= 1;
                if (lane == 0) { sb[wave][0] = bx0; sb[wave][1] = by0; sb[wave][2] = bx1; sb[wave][3] = by1; }
                __syncthreads();
                wx0 = INT_MAX; wy0 = INT_MAX; wx1 = INT_MIN; wy1 = INT_MIN;
#pragma unroll
                for (int w2 = 0; w2 < NW; ++w2) {
                    wx0 = min(wx0, sb[w2][0]); wy0 = min(wy0, sb[w2][1]);
                    wx1 = max(wx1, sb[w2][2]); wy1 = max(wy1, sb[w2][3]);
                }
                empty = wx0 > wx1;  // every sample of these planes is fully out of bounds
                if (empty) { wx0 = 0; wx1 = 0; wy0 = 0; wy1 = 0; }
                WC = ((wx1 - wx0 + 2) + 15) & ~15;  // +1 east tap, pitch multiple of 16
                WR = wy1 - wy0 + 2;                  // +1 south tap
                return WC * WR <= NTEX_MAX;   // block-uniform
            };
            // Smallest split into 1, 2 or 4 parts whose windows all fit LDS; the geometry is only redone
            // when the coarser split failed (large disparities per plane, e.g. 512x1024 with D=128).
            // (a group of only 8 planes starts at nsplit = 2: part 0 is then exactly planes k0 .. k0+7)
            int nsplit = kend - k0 <= 8 ? 2 : 1;
            for (;;) {
                geometry(nsplit);
                const int per = KP / nsplit;
                bool fits = true;
                for (int part = 0; part < nsplit && fits && k0 + part * NPG * per < kend; ++part)
                    fits = window_of(part * per, per);
                if (fits) break;
                if (nsplit == 4) {  // block-uniform: leave the tile (all its sub-tiles) to the gather kernel
                    if (lane == 0 && pgl == 0) flag_subtile();
                    goto tile_done;
                }
                nsplit *= 2;
            }
            const int per = KP / nsplit;

            for (int part = 0; part < nsplit && k0 + part * NPG * per < kend; ++part) {
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
                const int tl = sl * NT + tid;
                // floor(tl / WC): (tl + 0.5) / WC is at least 0.5 / WC away from an integer, far more than the
                // rounding error of the product for tl < 4096
                const int row = (int)(((float)tl + 0.5f) * __builtin_amdgcn_rcpf((float)WC)), col = tl - row * WC;
                const int gx = wx0 + col, gy = wy0 + row;
                const bool inb = !empty && row < WR && gx >= 0 && gx < a.W && gy >= 0 && gy < a.H;
                so[sl] = inb ? (gy * a.W + gx) * 16 : 0x7fffffff;
            }
            // reference features: wave w moves channel 4*ch + w of the tile's 64 pixels
            const int ro = p * 4;
            const unsigned win_lds = lds_addr_of(win + wave * 64), ref_lds = lds_addr_of(reft + (sub * NPG + pgl) * 64);
            auto stage = [&](int bufi, int ch) {
                const int soff = ch * HW * 16;
#pragma unroll
                for (int sl = 0; sl < SLOTS; ++sl)
                    if (sl * NT + wave * 64 < WC * WR)  // wave-uniform: this wave's 64 texels are part of the window
                        dma_b128(src_rsrc, win_lds + (bufi * NTEX_MAX + sl * NT) * 16, so[sl], soff);
                const int c = ch * 4 + pgl;
                // channels beyond C: out of range => zeros (the packed source is zero padded as well)
                dma_b32(ref_rsrc, ref_lds + bufi * (NSUB * 1024), c < aC ? ro : 0x7fffffff, c < aC ? c * HW * 4 : 0);
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
        const float* rp = reft + CUR * (NSUB * 256) + sub * 256 + lane;                             \
        const float4 rf = make_float4(rp[0], rp[64], rp[128], rp[192]);                             \
        if (per == 4) PDEPTH_COMPUTE(4, 0)                                                          \
        else if (per == 2) { if (part == 0) PDEPTH_COMPUTE(2, 0) else PDEPTH_COMPUTE(2, 2) }        \
        else if (part == 0) PDEPTH_COMPUTE(1, 0)                                                    \
        else if (part == 1) PDEPTH_COMPUTE(1, 1)                                                    \
        else if (part == 2) PDEPTH_COMPUTE(1, 2)                                                    \
        else PDEPTH_COMPUTE(1, 3)                                                                   \
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
                if (kpl[i] < kend) {
                    float* o = costs + (size_t)kpl[i] * 64 + lane;  // owned by this thread only
                    const float c = div_sigma(acc[i]);
                    *o = (v == 0) ? (0.0f + c) : (*o + c);
                }
            }
        }

        // ---- band group: planes [ks, D) in correlation form --------------------------------------------
        if (ks < aD) {  // block-uniform
            typedef const __attribute__((address_space(3))) v4f* lds_v4f;
            typedef const __attribute__((address_space(3))) float* lds_f;
            typedef __attribute__((address_space(3))) float* lds_fw;
            const int NX = NC * NR;
            const int WCW = gWC * gWR;  // <= BAND_TEX: one DMA instruction of waves 0..2 per chunk
            const int trow = (int)(((float)tid + 0.5f) * __builtin_amdgcn_rcpf((float)gWC)), tcol = tid - trow * gWC;
            const int tgx = gwx0 + tcol, tgy = gwy0 + trow;
            const bool tin = tid < WCW && tgx >= 0 && tgx < a.W && tgy >= 0 && tgy < a.H;
            const int so = tin ? (tgy * a.W + tgx) * 16 : 0x7fffffff;
            const bool has_win = wave * 64 < WCW;  // wave-uniform
            const int ro = p * 4;
            // Stage st = packed planes 2 st and 2 st + 1, except the last stage = the two Gram planes.
            const int gstage = (nchunk + 1) / 2, nstage = gstage + 1;
            const int shift = (BR - (nstage - 1) % BR) % BR;  // the last stage lands in ring slot 0
            auto stage = [&](int st) {
                const int q = (st + shift) % BR;
#pragma unroll
                for (int hlf = 0; hlf < 2; ++hlf) {
                    const int pl = st == gstage ? nchunk + hlf : 2 * st + hlf;  // packed plane
                    const bool feat = st < gstage && pl < nchunk;              // a channel group (else Gram / nothing)
                    if (has_win)
                        dma_b128(src_rsrc, win_lds0 + q * BAND_STAGE_BYTES + hlf * BAND_CHUNK_BYTES + wave * 1024,
                                 (feat || st == gstage) ? so : 0x7fffffff, pl * HW * 16);
                    const int c = pl * 4 + pgl;
                    const bool rok = feat && c < aC;
                    dma_b32(ref_rsrc, win_lds0 + BAND_REF_OFF + ((2 * st + hlf) % (2 * BR)) * BAND_REF_CHUNK + sub * 1024 + pgl * 256,
                            rok ? ro : 0x7fffffff, rok ? c * HW * 4 : 0);
                }
            };
            const int ndma = has_win ? 4 : 2;  // DMA instructions of this wave per stage
            // wave w accumulates the box texels j = w, w+4, w+8, ... (row-major in the NC x NR box)
            const int base = win_lds0 + ((bby0 - gwy0) * gWC + (bbx0 - gwx0)) * 16;  // this pixel's box, ring slot 0, chunk 0
            auto row_shift = [&](int r) { return ((r < 8 ? bsh0 : bsh1) >> ((r & 7) * 4)) & 15; };  // column offset of box row r
            int xaddr[XPW];
            float xacc[XPW];
            {
                int jr = 0, jc = pgl;
#pragma unroll
                for (int m = 0; m < XPW; ++m) {
                    while (jc >= NC) { jc -= NC; ++jr; }
                    xaddr[m] = (jr < NR ? base + (jr * gWC + jc + row_shift(jr)) * 16 : base);
                    xacc[m] = 0.0f;
                    jc += NPG;
                }
            }
            float rr = 0.0f;  // |r|^2 of this pixel
            lds_barrier();  // every reader of the previous group is done with the buffers
            for (int st = 0; st < BR - 1 && st < nstage; ++st) stage(st);
            for (int st = 0; st < nstage; ++st) {
                wait_dma_but(ndma * min(BR - 2, nstage - 1 - st));  // stage st has landed, younger ones stay in flight
                lds_barrier();
                if (st + BR - 1 < nstage) stage(st + BR - 1);  // into the slot of stage st-1
                if (st < gstage) {
                    const int qoff = ((st + shift) % BR) * BAND_STAGE_BYTES;
                    const float* rp0 = reinterpret_cast<const float*>(win) + BAND_REF_OFF / 4 +
                                       ((2 * st) % (2 * BR)) * (BAND_REF_CHUNK / 4) + sub * 256 + lane;
#pragma unroll
                    for (int hlf = 0; hlf < 2; ++hlf) {
                        if (2 * st + hlf < nchunk) {  // uniform
                            const float* rp = rp0 + hlf * (BAND_REF_CHUNK / 4);
                            const float4 rf = make_float4(rp[0], rp[64], rp[128], rp[192]);
                            rr = __builtin_fmaf(rf.x, rf.x, rr); rr = __builtin_fmaf(rf.y, rf.y, rr);
                            rr = __builtin_fmaf(rf.z, rf.z, rr); rr = __builtin_fmaf(rf.w, rf.w, rr);
#pragma unroll
                            for (int m0 = 0; m0 < XPW; m0 += 4) {
                                if (pgl + NPG * m0 < NX) {  // uniform: this group of four box texels exists
                                    v4f t[4];
#pragma unroll
                                    for (int u = 0; u < 4; ++u)
                                        t[u] = *(lds_v4f)(size_t)(unsigned)(xaddr[m0 + u] + qoff + hlf * BAND_CHUNK_BYTES);
#pragma unroll
                                    for (int u = 0; u < 4; ++u) {
                                        float x_ = xacc[m0 + u];
                                        x_ = __builtin_fmaf(t[u].x, rf.x, x_); x_ = __builtin_fmaf(t[u].y, rf.y, x_);
                                        x_ = __builtin_fmaf(t[u].z, rf.z, x_); x_ = __builtin_fmaf(t[u].w, rf.w, x_);
                                        xacc[m0 + u] = x_;
                                    }
                                }
                            }
                        }
                    }
                }
            }
            // Every stage has landed (the last wait was vmcnt(0)) and was published by the barrier of the last
            // iteration, which every wave reached only after its last channel stage: the Gram planes sit in ring
            // slot 0 and everything behind it is free for the X exchange buffer (slot j of every pixel at
            // xb + j*256 + lane*4).
            const int xb = win_lds0 + BAND_STAGE_BYTES + sub * (NX_MAX * 256) + lane * 4;
#pragma unroll
            for (int m = 0; m < XPW; ++m) {
                const int j = pgl + NPG * m;
                if (j < NX) *(lds_fw)(size_t)(unsigned)(xb + j * 256) = xacc[m];
            }
            lds_barrier();
            // combine: wave w takes planes ks + w, ks + w + 4, ...; Gram planes (N, H, V, D1) and (D2, -, -, -)
            const int g4b = win_lds0, g1b = g4b + BAND_CHUNK_BYTES;
            int viol = 0;
            // (wave-uniform: rectified stereo and the image axes of a forward motion have plain rectangles and skip the shifts)
            const bool sheared = PDEPTH_SHEAR && __builtin_amdgcn_ballot_w64((bsh0 | bsh1) != 0) != 0ull;
            for (int k = ks + pgl; k < aD; k += NPG) {
                float ix, iy;
                plane_sample_pos_fast(xf, t2a, t2b, t2c, dcl[k], cx, cy, rcx, rcy, half_w, half_h, ix, iy);
                // footprint as in make_footprint(); "any tap inside the image" <=> x0 in [-1, W-1] and y0 in [-1, H-1]
                const float xfl = floorf(ix), yfl = floorf(iy);
                float fw = ix - xfl, fe = 1.0f - fw, fn = iy - yfl, fs = 1.0f - fn;
                const int fx0 = (int)fminf(fmaxf(xfl, -2.0f), (float)(a.W + 1));
                const int fy0 = (int)fminf(fmaxf(yfl, -2.0f), (float)(a.H + 1));
                int dx = fx0 - bbx0, dy = fy0 - bby0;
                const bool any = live && ix == ix && iy == iy && (unsigned)(fx0 + 1) < (unsigned)(a.W + 1) &&
                                 (unsigned)(fy0 + 1) < (unsigned)(a.H + 1);
                if (!any) { fw = fw * 0.0f; fe = fe * 0.0f; fn = fn * 0.0f; fs = fs * 0.0f; dy = 0; }
                if ((unsigned)dy > (unsigned)(NR - 2)) { viol = 1; dy = 0; }
                // the two rows of the footprint start at their own column of the sheared box
                int dt, db;
                if (sheared) {
                    const unsigned shw = (unsigned)((((unsigned long long)(unsigned)bsh1 << 32) | (unsigned)bsh0) >> (dy * 4));  // one 64-bit shift
                    const int sh_t = (int)(shw & 15u), sh_b = (int)((shw >> 4) & 15u);
                    if (!any) dx = sh_t;
                    dt = dx - sh_t; db = dx - sh_b;
                    if (!any) db = 0;
                    if ((unsigned)dt > (unsigned)(NC - 2) || (unsigned)db > (unsigned)(NC - 2)) { viol = 1; dx = sh_t; dt = 0; db = 0; }
                } else {
                    if (!any) dx = 0;
                    if ((unsigned)dx > (unsigned)(NC - 2)) { viol = 1; dx = 0; }
                    dt = dx; db = dx;
                }
                const int slot = dy * NC + dt, slot_b = (dy + 1) * NC + db;
                const int tex = (((bby0 + dy - gwy0) * gWC + (bbx0 + dx - gwx0)) * 16);
                auto xat = [&](int j) { return *(lds_f)(size_t)(unsigned)(xb + j * 256); };
                const float X00 = xat(slot), X01 = xat(slot + 1), X10 = xat(slot_b), X11 = xat(slot_b + 1);
                const v4f G00 = *(lds_v4f)(size_t)(unsigned)(g4b + tex);
                const v4f G01 = *(lds_v4f)(size_t)(unsigned)(g4b + tex + 16);
                const v4f G10 = *(lds_v4f)(size_t)(unsigned)(g4b + tex + gWC * 16);
                const v4f G11 = *(lds_v4f)(size_t)(unsigned)(g4b + tex + gWC * 16 + 16);
                const float D2 = *(lds_f)(size_t)(unsigned)(g1b + tex);
                // |sum_t w_t s_t|^2, separable in the x weights (e, w) and the y weights (s, n)
                const float ee = fe * fe, ww = fw * fw, ew = fe * fw;
                const float A = ee * G00.x + ww * G01.x + 2.0f * ew * G00.y;   // top row:    N00, N01, H00
                const float B = ee * G10.x + ww * G11.x + 2.0f * ew * G10.y;   // bottom row: N10, N11, H10
                const float Cq = ee * G00.z + ww * G01.z + ew * (G00.w + D2);  // cross rows: V00, V01, D1 + D2
                const float Q = (fs * fs) * A + (fn * fn) * B + 2.0f * (fs * fn) * Cq;
                const float XW = (fs * fe) * X00 + (fs * fw) * X01 + (fn * fe) * X10 + (fn * fw) * X11;
                const float c = div_sigma((Q - 2.0f * XW) + rr);
                float* o = costs + (size_t)k * 64 + lane;
                *o = (v == 0) ? (0.0f + c) : (*o + c);
            }
            if (wave_max_s(viol) != 0 && lane == 0) flag_subtile();  // the gather kernel redoes the sub-tile
        }
    }
    __syncthreads();
    {
    // ---- epilogue from LDS: cost store, log-softmax over D, expectation ----------------------
    // wave w handles planes k = w, w+4, w+8, ... of the tile's 64 pixels
    float* const cost_out = PDEPTH_COLD_ARG(float*, cost_out);
    float* const logp_out = PDEPTH_COLD_ARG(float*, logp_out);
    float* const depth_out = PDEPTH_COLD_ARG(float*, depth_out);
    float* cout = (cost_out && live) ? cost_out + (size_t)b * aD * HW + p : nullptr;
    if (cout)
        for (int k = pgl; k < aD; k += NPG) cout[(size_t)k * HW] = costs[k * 64 + lane];
    if (logp_out || depth_out) {
        // Each wave reduces its planes locally (max, then sum of exp relative to its own max); ONE exchange of
        // (max, sum) per wave through the idle reference buffers, combined by rescaling -- one barrier instead of
        // two exchange rounds with two barriers each.
        float mw = -INFINITY;
        for (int k = pgl; k < aD; k += NPG) mw = fmaxf(mw, costs[k * 64 + lane]);
        // (exp_nonpos: 2^(x log2 e) on v_exp_f32 with an exact split of the product, ~1.5 ulp in 9 instructions against
        //  libm's 20; every argument here is <= 0.)
#define PDEPTH_EXPNP(x) exp_nonpos(x)
        float sw = 0.0f;
        for (int k = pgl; k < aD; k += NPG) sw = sw + PDEPTH_EXPNP(costs[k * 64 + lane] - mw);
        float* redm = reft + sub * 512;        // [NPG][64] of this sub-tile
        float* reds = reft + sub * 512 + 256;  // [NPG][64]
        redm[pgl * 64 + lane] = mw;
        reds[pgl * 64 + lane] = sw;
        __syncthreads();
        const float m0 = redm[lane], m1 = redm[64 + lane], m2 = redm[128 + lane], m3 = redm[192 + lane];
        const float m = fmaxf(fmaxf(m0, m1), fmaxf(m2, m3));
        // (a wave without planes -- D < 4 -- has max -inf and sum 0: exp(-inf - m) = 0, no NaN as long as m is finite;
        //  if every cost is -inf or NaN the result is NaN like the reference's)
        const float s = (reds[lane] * PDEPTH_EXPNP(m0 - m) + reds[64 + lane] * PDEPTH_EXPNP(m1 - m)) +
                        (reds[128 + lane] * PDEPTH_EXPNP(m2 - m) + reds[192 + lane] * PDEPTH_EXPNP(m3 - m));
        const float ls = logf(s);
        float e = 0.0f;
        float* o = (logp_out && live) ? logp_out + (size_t)b * aD * HW + p : nullptr;
        for (int k = pgl; k < aD; k += NPG) {
            const float lp = (costs[k * 64 + lane] - m) - ls;
            if (o) o[(size_t)k * HW] = lp;
            e = e + dcl[k] * PDEPTH_EXPNP(fminf(lp, 0.0f));
        }
        if (depth_out) {
            red[pgl * 64 + lane] = e;
            __syncthreads();
            if (pgl == 0 && live)
                depth_out[(size_t)b * HW + p] = (red[lane] + red[64 + lane]) + (red[128 + lane] + red[192 + lane]);
        }
    }
    }
tile_done:
    if (tid == 0) {   // publish the next item (the barrier below makes it visible)
        int nx = -1;
        if (queued) {
            if (!own_done && nxt_own < n_own) nx = nxt_own | (xcd << 24);
            else { own_done = true; nx = steal(); }
        }
        s_item[item_par ^ 1] = nx;
    }
    __syncthreads();  // the tile's LDS state is dead, s_item of the next round is visible
    item_par ^= 1;
    item = __builtin_amdgcn_readfirstlane(s_item[item_par]);   // (block-uniform)
    }  // work items
}

}  // namespace PDEPTH_VARIANT

static size_t tiled_lds_bytes(int D) {
    return (size_t)(NBUF * NTEX_MAX + NBUF * NSUB * 64) * sizeof(float4) + (size_t)NSUB * (D + NPG) * 64 * sizeof(float) +
           (size_t)(D + 2 * (D / 8 + 1)) * sizeof(float);
}

// Largest D whose cost tile fits LDS next to the window (2 blocks per CU).

#if PDEPTH_NSUB == 1
// Largest D whose cost tile fits LDS next to the window buffers (2 blocks per CU).
int sweep_tiled_max_planes() { return 160; }
#endif

static size_t flag_only_bytes(int B, int H, int W) { return sweep_ws_flag_only_bytes(B, H, W); }
static size_t flag_bytes(int B, int H, int W) { return sweep_ws_flag_bytes(B, H, W); }
#if PDEPTH_NSUB == 1
// Two tiles per block pay off when two such blocks fit a CU (the cost tiles of both sub-tiles live in LDS: D <= 64)
// and the image is large enough for the wider windows not to dominate; measured on the BASELINE configurations.
// (PDEPTH_ALGO_TILED_1 / _2 force a variant.)
hipError_t launch_sweep_tiled(const SweepArgs& a, void* workspace, hipStream_t stream, bool packed_ready, int phases) {
    const bool two = a.D <= 64 && (long long)a.H * a.W >= 96 * 1024 && a.W >= 128;
    return two ? launch_sweep_tiled_n2(a, workspace, stream, packed_ready, phases) : launch_sweep_tiled_n1(a, workspace, stream, packed_ready, phases);
}

#endif

// Launches the pre-pass, this variant's tiled kernel, then the gather kernel on the tiles it flagged.
// lab builds, PDEPTH_NO_SPEC=1 (read once): always the general instantiation (A/B timing of the compile-time specialisation)
static bool getenv_once_no_spec() {
#ifdef PDEPTH_LAB
    static const bool v = [] { const char* e = getenv("PDEPTH_NO_SPEC"); return e && e[0] == '1'; }();
    return v;
#else
    return false;
#endif
}

hipError_t PDEPTH_CAT(launch_sweep_tiled_n, PDEPTH_NSUB)(const SweepArgs& a, void* workspace, hipStream_t stream, bool packed_ready, int phases) {
    const int tiles16_x = (a.W + TW - 1) / TW, tiles_y = (a.H + TH - 1) / TH;  // the gather kernel's (and the flags') tiles
    const int tiles_x = (a.W + TW * NSUB - 1) / (TW * NSUB);                  // this kernel's work items per row
    const int tiles = tiles_x * tiles_y;
    int* flags = reinterpret_cast<int*>(workspace);
    float4* packed = reinterpret_cast<float4*>(static_cast<char*>(workspace) + flag_bytes(a.B, a.H, a.W));
    hipError_t e = hipSuccess;
    if (phases & PH_PRE) e = packed_ready ? clear_sweep_flags(a, workspace, stream) : launch_pack_c4(a, workspace, stream, /*centre=*/false);
    if (e != hipSuccess) return e;
    const size_t lds = tiled_lds_bytes(a.D);
    int* queue = reinterpret_cast<int*>(static_cast<char*>(workspace) + flag_only_bytes(a.B, a.H, a.W));
    // persistent grid: as many blocks as the device holds at once (3 per CU at D <= 64), a multiple of 8
    const int n_cu = sweep_device_cus();
    const int per_cu = (int)((160 * 1024) / (lds + 640));
    constexpr int max_per_cu = PDEPTH_OCC * 4 / NW;  // blocks per CU the register budget allows
    int nblk = n_cu * (per_cu < 1 ? 1 : per_cu > max_per_cu ? max_per_cu : per_cu);
    nblk = (nblk + 7) & ~7;
    const long long full = 8ll * ((tiles + 7) / 8) * a.B;  // one block per item of the largest XCD band, times 8
    if (full <= nblk) nblk = (int)full;
    dim3 grid(nblk);
    // (routing: the pre-pass / the flag clear of this call has set flag 1 of the statistics row of every ill-conditioned item)
    const float* route_stats = reinterpret_cast<const float*>(static_cast<char*>(workspace) + sweep_ws_stats_offset(a.B, a.V, a.C, a.H, a.W));
    // (the dynamic-LDS attribute is per kernel, sticky and the same on every device: set it whenever more than the
    //  default is needed -- no cached state, and a failure is reported instead of surfacing as a launch error)
    if (phases & PH_KERNEL) {
    if (a.metric == 0 && a.D == 64 && a.C == 67 && a.V == 1 && !getenv_once_no_spec()) {
        auto kern = PDEPTH_VARIANT::sweep_tiled_kernel<0, true>;
        if (lds > 64 * 1024) {
            e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
        }
        hipLaunchKernelGGL(kern, grid, dim3(NT), lds, stream, a, packed, flags, queue, tiles_x, tiles, route_stats);
    } else if (a.metric == 0) {
        auto kern = PDEPTH_VARIANT::sweep_tiled_kernel<0, false>;
        if (lds > 64 * 1024) {
            e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
        }
        hipLaunchKernelGGL(kern, grid, dim3(NT), lds, stream, a, packed, flags, queue, tiles_x, tiles, route_stats);
    } else {
        auto kern = PDEPTH_VARIANT::sweep_tiled_kernel<1, false>;
        if (lds > 64 * 1024) {
            e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
        }
        hipLaunchKernelGGL(kern, grid, dim3(NT), lds, stream, a, packed, flags, queue, tiles_x, tiles, route_stats);
    }
    }
    e = hipGetLastError();
    if (e != hipSuccess || !(phases & PH_GATHER)) return e;
    SweepArgs ag = a;
    ag.packed_src = packed;   // (the gather kernel's source when the caller passed a packed source only)
    return launch_sweep_direct_flagged(ag, flags, queue + GATHER_COUNT_SLOT, tiles16_x, tiles16_x * tiles_y, stream);
}

}  // namespace pdepth
