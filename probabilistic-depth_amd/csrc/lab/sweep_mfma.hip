// Plane sweep with the channel contraction on the matrix pipe (L2 metric): the default path of pdepth_sweep_{cost,dpv}_f32
// for the feature widths it is instantiated for.
//
// Every plane is evaluated in the correlation form of the L2 distance (the band mode of sweep_tiled.hip, here for all
// planes):
//       sum_c (sum_t w_t s_t[c] - r[c])^2 = w^T G w - 2 sum_t w_t X_t + |r|^2,        X_t = <r, s_t>,
// with G (Gram terms of neighbouring source texels) from the pre-pass (pack_c4_kernel, sweep_tiled.hip).  What is left
// of the channel loop is X: one dot product per (pixel, source texel the pixel's planes touch).  For 16 neighbouring
// reference pixels those texels overlap almost completely -- the epipolar segments of neighbours are translates of each
// other -- so X for 16 pixels x 16 consecutive texels of a source row is a 16x16xC matrix product, and the fp32 matrix
// instruction (v_mfma_f32_16x16x4_f32: exact fp32 products, fp32 accumulation, the vector unit's peak rate on a pipe of
// its own) computes it from two registers per lane and 4 channels, with no LDS traffic per multiply.  About half of the
// products are for (pixel, texel) pairs no plane samples; they cost matrix-pipe time only, which the kernel has to spare.
//
//   work item = one 16x4 tile of reference pixels of one batch item; a workgroup = 4 waves = the tile's four sub-blocks of
//               16 pixels (16x1 or 8x2, one decision per batch item).  The waves share nothing but the caches -- they run
//               side by side so that they find each other's source lines in L1 -- and meet at one raw barrier per tile;
//               persistent workgroups pull tiles from per-XCD queues (balanced partition, one tile of look-ahead);
//   lanes     = in the vector phases lane (px = lane & 15, pg = lane >> 4) owns pixel px and planes k = 4 j + pg
//               (interleaved, so any range of j keeps all lanes busy); in the matrix phase lane (n, kq) feeds texel /
//               pixel n and channel slice kq;
//   geometry  = bit-faithful sample positions (geometry.hpp), once per (pixel, plane, view), kept in registers;
//   row table = for the planes of a pass: per source row the run [lo, hi] of texels any of the 16 pixels touches
//               (LDS min/max, each lane folding its consecutive planes of equal row first), cut into blocks of 16 texels;
//               a pass = a range of j whose runs need at most MAXB blocks: all 64 planes where that fits, otherwise the
//               range is halved (near planes, whose segments are long, end up in passes of their own);
//   X         = per block: texel features by buffer_load_dwordx4 from the packed source (lane (n, kq) loads 4 channels of
//               texel n: out-of-image texels are fetched out of range and arrive as zeros = padding_mode 'zeros'),
//               reference features of the 16 pixels held in registers for the whole sub-block, ceil(C/4) MFMAs in two
//               accumulator chains, three blocks' loads in flight; result X[texel][pixel] to LDS (16 pixels x 16 MAXB slots), the Gram
//               records of the block's texels next to it;
//   combine   = per (pixel, plane): 4 X values, the Gram terms of the cell and the bilinear weights, cost accumulated
//               over views in registers;
//   epilogue  = cost store, log-softmax over D and E[d] from registers (4 lanes per pixel, two xor-shuffles).
//
// A tile whose geometry does not fit (a single j -- 4 planes -- needing more than MAXB blocks or 64 source rows: extreme
// poses) is flagged and redone by the gather kernel of sweep_direct.hip, like the tiles the other kernels hand over.
// Agreement with the gather kernel (reference op order): an ulp or two of the largest cost of the volume, as the band mode.
#include <hip/hip_runtime.h>

#include <climits>

#include "../geometry.hpp"
#include "../kernels.hpp"
#include "../pick.hpp"

namespace pdepth {

namespace {

typedef float v4f __attribute__((ext_vector_type(4)));
typedef int v4i __attribute__((ext_vector_type(4)));

#ifndef MFMA_MAXB
#define MFMA_MAXB 15
#endif
#ifndef MFMA_WG_PER_CU
#define MFMA_WG_PER_CU 2
#endif
#ifndef MFMA_OCC   // waves per SIMD the register allocation must allow (experiment: 3 with MFMA_MAXB=9 and MFMA_WG_PER_CU=3)
#define MFMA_OCC 2
#endif
#ifndef MFMA_ABL_STOP   // timing experiments (results wrong): 1 = skip everything after the sample positions, 2 = after the row
#define MFMA_ABL_STOP 0 // table, 3 = after X, 4 = after the combine (no epilogue)
#endif
constexpr int MAXB = MFMA_MAXB;        // blocks of 16 texels per pass (LDS: 4 waves x 19.9 KB, two workgroups per CU)
constexpr int XSTRIDE = MAXB * 16 + 4; // floats per pixel of the X buffer (stride/4 odd: conflict-free b128 stores)
constexpr int MAXROWS = 64;            // source rows per pass (one lane per row)
constexpr int OOB = 0x7fffffff;        // buffer offset beyond every descriptor: the load returns 0

__device__ __forceinline__ int opaque_v(int x) { asm volatile("" : "+v"(x)); return x; }
__device__ __forceinline__ int opaque_s(int x) { asm volatile("" : "+s"(x)); return x; }
__device__ __forceinline__ float opaque_f(float x) { asm volatile("" : "+v"(x)); return x; }

// Diagnostic build only (-DMFMA_STAMPS, tools/dbg/mfma_stamps.py): cycles per phase and event counts, summed over waves,
// in the spare ints behind the queue counters.  No stamp exists in the product build.
#ifdef MFMA_STAMPS
#define MSTAMP(idx) { const unsigned long long now_ = __builtin_amdgcn_s_memrealtime(); stamp_acc[idx] += now_ - stamp_t; stamp_t = now_; }
#define MCOUNT(idx, v) { stamp_acc[idx] += (unsigned long long)(v); }
#elif defined(MFMA_MARKS)   // phase markers in the ISA listing (tools/dbg/isa_phases.py): comments only
#define MSTAMP(idx) asm volatile("; MARK " #idx);
#define MCOUNT(idx, v)
#else
#define MSTAMP(idx)
#define MCOUNT(idx, v)
#endif

// The waves of a workgroup share nothing: what a wave writes to its LDS region only its own lanes read.  LDS operations of
// one wave complete in order, so "every earlier LDS operation of this wave is done" is all the synchronisation there is
// (no s_barrier, and unlike __syncthreads() no wait for the global loads in flight).
#define WAVE_LDS_SYNC() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

// inclusive prefix sum over the 64 lanes: Hillis-Steele inside the rows of 16 (DPP row_shr), then the row totals
__device__ __forceinline__ int wave_scan_incl(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);   // row_bcast:15 into rows 1 and 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);   // row_bcast:31 into rows 2 and 3
    return v;
}

#define MFMA_DPP_STEP(OP, ctrl) v = OP(v, __builtin_amdgcn_update_dpp(v, v, ctrl, 0xf, 0xf, false))
__device__ __forceinline__ int wave_min_i(int v) {
    MFMA_DPP_STEP(min, 0xB1); MFMA_DPP_STEP(min, 0x4E); MFMA_DPP_STEP(min, 0x141); MFMA_DPP_STEP(min, 0x140);
    return min(min(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)),
               min(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}
__device__ __forceinline__ int wave_max_i(int v) {
    MFMA_DPP_STEP(max, 0xB1); MFMA_DPP_STEP(max, 0x4E); MFMA_DPP_STEP(max, 0x141); MFMA_DPP_STEP(max, 0x140);
    return max(max(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)),
               max(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}
#undef MFMA_DPP_STEP

// plane_sample_pos_fast() of geometry.hpp for TWO planes at a time in packed fp32 (v_pk_mul / v_pk_add / v_pk_fma_f32: each
// component rounds exactly like the scalar instruction, so the positions are bit-identical): the kernel runs at two waves
// per SIMD, where the instruction rate of a wave, not the lane rate, is what the vector phases are short of.
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f splat2(float x) { return v2f{x, x}; }
__device__ __forceinline__ v2f fma2(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ v2f div_core2(v2f n, v2f d, v2f y) {
    const v2f q0 = n * y;
    const v2f r0 = fma2(-d, q0, n);
    const v2f q1 = fma2(r0, y, q0);
    const v2f r1 = fma2(-d, q1, n);
    return fma2(r1, y, q1);
}
__device__ __forceinline__ void plane_sample_pos_fast2(const ViewXform& x, float t2a, float t2b, float t2c, v2f d, float cx, float cy,
                                                       float rcx, float rcy, float half_w, float half_h, v2f& ix, v2f& iy) {
    const v2f px = splat2(x.kt[0]) + splat2(t2a) * d;
    const v2f py = splat2(x.kt[1]) + splat2(t2b) * d;
    const v2f pz = splat2(x.kt[2]) + splat2(t2c) * d;
    const v2f den = pz + splat2(1e-10f);
    const v2f y0 = v2f{__builtin_amdgcn_rcpf(den.x), __builtin_amdgcn_rcpf(den.y)};
    const v2f e = fma2(-den, y0, splat2(1.0f));
    const v2f y = fma2(e, y0, y0);
    const v2f u = div_core2(px, den, y);
    const v2f v = div_core2(py, den, y);
    const v2f gx = div_core2(u - splat2(cx), splat2(cx), splat2(rcx));
    const v2f gy = div_core2(v - splat2(cy), splat2(cy), splat2(rcy));
    ix = fma2(gx + splat2(1.0f), splat2(half_w), splat2(-0.5f));
    iy = fma2(gy + splat2(1.0f), splat2(half_h), splat2(-0.5f));
}

// Footprint of a sample position as make_footprint() (geometry.hpp) computes it, packed: (y0 << 16) | (x0 & 0xffff) of the
// top-left texel, or NO_CELL when no tap lies inside the image (NaN positions included); fw, fn = the fractions.
constexpr int NO_CELL = INT_MIN;
__device__ __forceinline__ int cell_of(float ix, float iy, int W, int H, float& fw, float& fn) {
    const float xfl = floorf(ix), yfl = floorf(iy);
    fw = ix - xfl;
    fn = iy - yfl;
    const int x0 = (int)fminf(fmaxf(xfl, -2.0f), (float)(W + 1));
    const int y0 = (int)fminf(fmaxf(yfl, -2.0f), (float)(H + 1));
    const bool any = ix == ix && iy == iy && (unsigned)(x0 + 1) < (unsigned)(W + 1) && (unsigned)(y0 + 1) < (unsigned)(H + 1);
    return any ? (y0 << 16) | (x0 & 0xffff) : NO_CELL;
}
__device__ __forceinline__ int cell_x(int xy) { return (int)(short)(xy & 0xffff); }
__device__ __forceinline__ int cell_y(int xy) { return xy >> 16; }

// LDS of one wave
struct __attribute__((aligned(16))) WaveLds {
    float Xs[16 * XSTRIDE];     // X[pixel][slot]
    float G4s[MAXB * 16 * 4];   // Gram record (N, H, V, D1 + D2) per slot
    int cmin[MAXROWS], cmax[MAXROWS];   // per cell row: min / max x0
    int rowoff[MAXROWS];        // per texel row: slot = x + rowoff
    int blk[MAXB + 8];          // per block: (y << 16) | (x & 0xffff) of its first texel
};

// NPL = packed feature planes of a source view (ceil(C / 4)); NHALF = ceil(D / 64).
template <int NPL, int NHALF>
__global__ __launch_bounds__(256, MFMA_OCC) void sweep_mfma_kernel(SweepArgs a, const float4* __restrict__ packed,
                                                             int* __restrict__ tile_flags, int* __restrict__ queue, int tiles_x,
                                                             int ntile) {
    constexpr int NCH = NPL / 4, NTL = NPL % 4;   // chunks of 16 channels (4 MFMAs per 16-byte load), left-over planes of 4
    __shared__ WaveLds wlds[4];
    __shared__ float dcl[64 * NHALF];

    if (a.pick == PICK_RUN_IF_SET && queue[PICK_SLOT] == 0) return;   // (the pre-pass chose the other kernel: pick.hpp)
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    WaveLds& L = wlds[wave];
    float* const Xs = L.Xs; float* const G4s = L.G4s;
    int* const cmin = L.cmin; int* const cmax = L.cmax; int* const rowoff = L.rowoff; int* const blk = L.blk;
    const int D = a.D, H = a.H, W = a.W, V = a.V, C = a.C;

    for (int k = threadIdx.x; k < 64 * NHALF; k += 256) dcl[k] = a.d_candi[min(k, D - 1)];
    __syncthreads();   // (the only barrier of the kernel: from here on the four waves go their own way)

    const float half_w = (float)W / 2.0f, half_h = (float)H / 2.0f;
    const float sigma = a.sigma, rsigma = refined_rcp(a.sigma);
    // v / sigma through the divide chain of geometry.hpp (bit-identical to the IEEE divide for finite operands in range);
    // inf / NaN (non-finite features) propagate through the plain product
    auto div_sigma = [&](float v) { return fabsf(v) < 1.0e30f ? div_core(v, sigma, rsigma) : v * rsigma; };
#ifdef MFMA_STAMPS
    unsigned long long stamp_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long stamp_t = __builtin_amdgcn_s_memrealtime();
#endif
#ifdef MFMA_EXIT_STAMPS
    const unsigned long long exit_t0 = __builtin_amdgcn_s_memrealtime();
#endif

    // Shape of a tile's four pixel sub-blocks (16 pixels each, one per wave): 16x1 (pixel rows) where the epipolar lines of
    // view 0 run along the source rows (rectified stereo: a row's 16 pixels share two source rows), else 8x2 (the texels of
    // two half rows overlap more than those of one full row: 30 % fewer blocks on a forward motion).  One decision per
    // batch item, from four probe pixels: does the sample move by more than half a source row between the first and the
    // last plane?  (Any choice is correct; this one is within 1 % of the best choice per tile on both benchmark poses.)
    __shared__ unsigned char s_wide[64];
    // (thread = (batch item, probe): the four probes of an item side by side, not one after the other -- on a small problem
    //  the prologue is a visible part of the launch)
    if (threadIdx.x < 64) s_wide[threadIdx.x] = 1;
    __syncthreads();
    if ((int)(threadIdx.x >> 2) < min(a.B, 64) && epipolar_probe_is_steep(a, threadIdx.x >> 2, threadIdx.x & 3)) s_wide[threadIdx.x >> 2] = 0;

    // Persistent workgroups: the grid fills the chip once (two workgroups per CU) and every workgroup pulls tiles -- (batch
    // item, 16x4 tile) -- from the queue of its XCD; wave w takes the tile's sub-block w, so the four waves, which read
    // mostly the same source texels, run side by side on one CU and find each other's lines in its L1.  XCD-aware
    // (workgroups are dealt round-robin over the 8 XCDs): every XCD owns one contiguous band of tiles, walked column by
    // column where the band is a whole number of tile rows, so that overlapping source texels hit that XCD's L2.  A
    // workgroup whose own queue is exhausted takes tiles of the other queues: the bands do not cost the same (on a forward
    // motion those at the top and bottom of the image see the longest epipolar segments).
    __shared__ int s_item[2];
    const int xcd = blockIdx.x & 7, qq = ntile >> 3, rr8 = ntile & 7;
    auto band_tiles_of = [&](int q) { return qq + (q < rr8 ? 1 : 0); };
    auto band_first_of = [&](int q) { return q < rr8 ? q * (qq + 1) : rr8 * (qq + 1) + (q - rr8) * qq; };
    const bool colmajor = rr8 == 0 && qq % tiles_x == 0;
    // Queue protocol (thread 0): the atomic on the workgroup's OWN queue is issued at the top of a tile and its result is
    // looked at when the tile is done -- a round trip to L2 that nothing waits for.  Only when the own band is exhausted (the
    // end of a launch) the other queues are polled, synchronously.
    const int n_own = band_tiles_of(xcd) * a.B;
    bool own_done = false;
    auto steal = [&]() -> int {   // (queue << 28) | index in the queue, or -1: every queue is exhausted
        for (int j = 1; j < 8; ++j) {
            const int q = (xcd + j) & 7, nq = band_tiles_of(q) * a.B;
            if (*(volatile int*)&queue[q] >= nq) continue;
            const int got = atomicAdd(&queue[q], 1);
            if (got < nq) return (q << 28) | got;
        }
        return -1;
    };
    auto resolve = [&](int got) -> int {   // result of the own-queue atomic -> item
        if (!own_done && got < n_own) return (xcd << 28) | got;
        own_done = true;
        return steal();
    };
    // item -> batch item, tile
    // floor(nn / dd) for 0 <= nn < 2^22, 0 < dd: (nn + 0.5) / dd is at least 0.5 / dd away from an integer, far more than the
    // rounding error of the product with the reciprocal (the integer divide costs ~40 dependent instructions, four per tile)
    auto fdiv = [](int nn, int dd) { return (int)(((float)nn + 0.5f) * __builtin_amdgcn_rcpf((float)dd)); };
    const bool small_idx = (long long)ntile * a.B < (1ll << 22);
    const int tiles_y_ = (H + 3) / 4;
    const bool balanced = rr8 == 0 && tiles_y_ % 16 == 0 && tiles_y_ * tiles_x == ntile;
    auto decode = [&](int item, int& b_, int& tx_, int& ty_) {
        const int q_ = item >> 28, iq = item & 0x0fffffff, band_tiles = band_tiles_of(q_);
        b_ = small_idx ? fdiv(iq, band_tiles) : iq / band_tiles;
        const int ti = iq - b_ * band_tiles;
        int tile = band_first_of(q_) + ti;
#ifndef MFMA_NO_BALANCE
        if (balanced) {
            // XCD q owns half-bands q and 8 + q of the image's 16 (q = 0: the top strip and the strip just below the middle):
            // on a forward motion the cost of a tile grows with its distance from the image centre, and this way every XCD
            // gets the same mix; the heavier half first and, inside a half, columns from both image borders inwards -- the
            // launch ends on cheap tiles.  (Any static partition is valid: queues that run dry steal from the others.)
            const int hb_rows = tiles_y_ / 16, half_tiles = hb_rows * tiles_x;
            const int second = ti >= half_tiles ? 1 : 0, tih = ti - second * half_tiles;
            const int hbi = (q_ < 4) == (second == 0) ? q_ : 8 + q_;
            const int cc = small_idx ? fdiv(tih, hb_rows) : tih / hb_rows, r_ = tih - cc * hb_rows;
            const int col = (cc & 1) ? tiles_x - 1 - (cc >> 1) : (cc >> 1);
            tile = (hbi * hb_rows + r_) * tiles_x + col;
        } else
#endif
        if (colmajor) {
            const int band_rows = qq / tiles_x, tc = small_idx ? fdiv(ti, band_rows) : ti / band_rows;
            tile = (q_ * band_rows + (ti - tc * band_rows)) * tiles_x + tc;
        }
        ty_ = small_idx ? fdiv(tile, tiles_x) : tile / tiles_x;
        tx_ = tile - ty_ * tiles_x;
    };
    // this wave's pixel of a tile (sub-block = wave) and the loads that do not depend on anything computed: the pixel's ray
    // and its reference features -- lane (n, kq): pixel n, channel slice kq, the B operands of every MFMA of the sub-block.
    // They are issued one tile ahead, so that their latency (first touch: HBM) lies under the previous tile's work.
    auto pixel_of = [&](int b_, int tx_, int ty_, int n_, int& x_, int& y_) {
        const bool wide_ = b_ < 64 ? s_wide[b_] != 0 : false;
        x_ = wide_ ? tx_ * 16 + n_ : tx_ * 16 + 8 * (wave & 1) + (n_ & 7);
        y_ = wide_ ? ty_ * 4 + wave : ty_ * 4 + 2 * (wave >> 1) + (n_ >> 3);
    };
    auto load_pixel = [&](int item, float(&Rr_)[NPL], float(&ray_)[3]) {
        int b_, tx_, ty_, x_, y_;
        decode(item, b_, tx_, ty_);
        const int n_ = opaque_v(lane & 15), kq_ = opaque_v(lane >> 4), HW_ = opaque_s(H * W);
        pixel_of(b_, tx_, ty_, n_, x_, y_);
        const int p_ = min(y_, H - 1) * W + min(x_, W - 1);
        // (buffer loads: 32-bit offsets, the channel as the scalar offset; channels beyond C are fetched out of range = 0)
        const __amdgpu_buffer_rsrc_t rray = __builtin_amdgcn_make_buffer_rsrc((void*)(a.rays + (size_t)b_ * 3 * HW_), 0, 3 * HW_ * 4, 0x00020000);
#pragma unroll
        for (int i = 0; i < 3; ++i) ray_[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rray, p_ * 4, i * HW_ * 4, 0));
        const __amdgpu_buffer_rsrc_t rref =
            __builtin_amdgcn_make_buffer_rsrc((void*)(a.ref + (size_t)b_ * a.ref_bstride), 0, C * HW_ * 4, 0x00020000);
#pragma unroll
        for (int g = 0; g < NCH; ++g)
#pragma unroll
            for (int i = 0; i < 4; ++i)
                Rr_[4 * g + i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                    rref, 16 * g + 4 * kq_ + i < C ? (4 * kq_ * HW_ + p_) * 4 : OOB, (16 * g + i) * HW_ * 4, 0));
#pragma unroll
        for (int tp = 0; tp < NTL; ++tp)
            Rr_[4 * NCH + tp] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                rref, 4 * (4 * NCH + tp) + kq_ < C ? (kq_ * HW_ + p_) * 4 : OOB, 4 * (4 * NCH + tp) * HW_ * 4, 0));
    };

    // s_item: this tile and the next.  (One tile of look-ahead and no more: a tile fetched early is a tile no other
    // workgroup can take -- with two tiles of look-ahead the workgroups ran dry over the last 20 % of a launch, each
    // finishing the tiles it had put aside while its neighbours idled.)
    // (when the grid covers every tile -- small problems -- workgroup i simply takes tile i of its XCD's band: no atomics on
    //  the critical path of a launch that is latency bound anyway)
    const bool one_each = (long long)gridDim.x >= 8ll * ((ntile + 7) / 8) * a.B;
    if (threadIdx.x == 0) {
        if (one_each) s_item[0] = (int)(blockIdx.x >> 3) < n_own ? (xcd << 28) | (int)(blockIdx.x >> 3) : -1;
        else s_item[0] = resolve(atomicAdd(&queue[xcd], 1));
    }
    __syncthreads();
    int slot = 0;
    float Rr[NPL], ray[3];
    int item = __builtin_amdgcn_readfirstlane(s_item[0]);
    if (item >= 0) load_pixel(item, Rr, ray);
    while (item >= 0) {
    int got_own = 0;
    if (threadIdx.x == 0 && !one_each && !own_done) got_own = atomicAdd(&queue[xcd], 1);   // (the next tile: issued now, looked at when this one is done)
    const int sub = wave;
    int b, tx, ty;
    decode(item, b, tx, ty);
    const float cx = a.cxcy[b * 2 + 0], cy = a.cxcy[b * 2 + 1];
    const float rcx = refined_rcp(cx), rcy = refined_rcp(cy);
    bool failed = false;   // wave-uniform
    const bool wide = b < 64 ? s_wide[b] != 0 : false;

    if ((wide ? ty * 4 + sub : ty * 4 + 2 * (sub >> 1)) < H) {   // (else: the sub-block lies below the image)
        // (opaque: the optimiser otherwise hoists every lane- and HW-derived invariant of the phases -- masks, LDS addresses,
        //  scalar offsets -- to the top of the kernel and spills them)
        const int n = opaque_v(lane & 15), kq = opaque_v(lane >> 4), HW = opaque_s(H * W);
        int x, y;
        pixel_of(b, tx, ty, n, x, y);
        const bool xlive = x < W && y < H;
        const int p = min(y, H - 1) * W + min(x, W - 1);
        const float r0 = ray[0], r1 = ray[1], r2 = ray[2];
        float rr = 0.0f;   // |r|^2 of the pixel
#pragma unroll
        for (int i = 0; i < NPL; ++i) rr = __builtin_fmaf(Rr[i], Rr[i], rr);
        rr = rr + __shfl_xor(rr, 16);
        rr = rr + __shfl_xor(rr, 32);

        float cost[NHALF * 16];
#pragma unroll
        for (int i = 0; i < NHALF * 16; ++i) cost[i] = 0.0f;
        MSTAMP(0)   // row setup: rays, reference features

        if (MFMA_ABL_STOP == 9) { cost[0] = rr + r0 + r1 + r2; }
        for (int v = 0; v < (MFMA_ABL_STOP == 9 ? 0 : V) && !failed; ++v) {
            ViewXform xf;
            make_view_xform(a.K + b * 9, a.R + ((size_t)b * V + v) * 9, a.t + ((size_t)b * V + v) * 3, a.blas_mode, xf);
            float t2a, t2b, t2c;
            ray_term2(xf, r0, r1, r2, t2a, t2b, t2c);
            const float4* srcv = packed + ((size_t)b * V + v) * (NPL + 2) * HW;
            const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)srcv, 0, (NPL + 2) * HW * 16, 0x00020000);

#pragma unroll
            for (int h = 0; h < NHALF; ++h) {
                if (failed) break;
                // ---- sample positions of this lane's 16 planes (NaN: plane beyond D or pixel beyond the image) ----------
                int xy[16];
                float fw[16], fn[16];
#ifdef MFMA_REP_GEOM   // timing experiment: the phase twice (same results) -- its cost in situ is the difference
                for (int rep = 0; rep < 2; ++rep) {
                t2a = opaque_f(t2a);
#endif
#pragma unroll
                for (int j = 0; j < 16; j += 2) {
                    const int k = 64 * h + 4 * j + kq;
                    v2f ix, iy;
                    plane_sample_pos_fast2(xf, t2a, t2b, t2c, v2f{dcl[k], dcl[k + 4]}, cx, cy, rcx, rcy, half_w, half_h, ix, iy);
                    xy[j] = cell_of(ix.x, iy.x, W, H, fw[j], fn[j]);
                    xy[j + 1] = cell_of(ix.y, iy.y, W, H, fw[j + 1], fn[j + 1]);
                    if (k >= D || !xlive) xy[j] = NO_CELL;
                    if (k + 4 >= D || !xlive) xy[j + 1] = NO_CELL;
                    if ((j & 6) == 6) __builtin_amdgcn_sched_barrier(0);   // (four pairs' chains at a time)
                }
#ifdef MFMA_REP_GEOM
                }
#endif

                MSTAMP(1)   // sample positions
                int j0 = MFMA_ABL_STOP == 1 ? 16 : 0, len = 16;   // current pass: j in [j0, j0 + len)
                if (MFMA_ABL_STOP == 1) { for (int j = 0; j < 16; ++j) cost[h * 16 + j] += fw[j] + fn[j] + (float)xy[j]; }
                while (j0 < 16) {
                    const int j1 = j0 + len;
                    // (pinned per pass and per phase: nothing derived from the cells is carried from one phase into the next)
#pragma unroll
                    for (int j = 0; j < 16; ++j) xy[j] = opaque_v(xy[j]);
                    // ---- row table -----------------------------------------------------------------------------
#ifdef MFMA_REP_TABLE
                    int nb = 0, ybase = 0;
                    bool fits = true;
                    for (int rep = 0; rep < 2; ++rep) {
                    nb = 0; fits = true;
#endif
                    // the cells of the pass (planes outside [j0, j1): none), so that the loops below run over all 16 without
                    // range branches
                    int xyp[16];
#pragma unroll
                    for (int j = 0; j < 16; ++j) xyp[j] = (j >= j0 && j < j1) ? xy[j] : NO_CELL;
                    int lmin = INT_MAX, lmax = INT_MIN;
#pragma unroll
                    for (int j = 0; j < 16; ++j) {
                        const bool valid = xyp[j] != NO_CELL;
                        lmin = min(lmin, valid ? cell_y(xyp[j]) : INT_MAX);
                        lmax = max(lmax, valid ? cell_y(xyp[j]) : INT_MIN);
                    }
#ifdef MFMA_REP_TABLE
                    ybase = wave_min_i(lmin);
                    const int ytop = wave_max_i(lmax);
#else
                    const int ybase = wave_min_i(lmin), ytop = wave_max_i(lmax);
                    int nb = 0;
                    bool fits = true;
#endif
                    if (ybase <= ytop) {
                        const int ncell = ytop - ybase + 1;
                        if (ncell + 1 > MAXB) {   // (every texel row takes a block: cannot fit, and MAXB < MAXROWS)
                            fits = false;
                        } else {
                            cmin[lane] = INT_MAX;
                            cmax[lane] = INT_MIN;
                            WAVE_LDS_SYNC();
                            // per cell row the min / max x0 over the wave by LDS atomics; a lane first folds its consecutive planes
                            // of equal row (the position moves monotonically along the epipolar line), so that on a rectified pair
                            // -- every plane of every pixel in one row -- a lane issues one pair of atomics, not sixteen that
                            // serialise on one address
                            int run = -1, rmin = 0, rmax = 0;
#pragma unroll
                            for (int j = 0; j < 16; ++j) {
                                if (xyp[j] != NO_CELL) {
                                    const int r = cell_y(xyp[j]) - ybase, cxx = cell_x(xyp[j]);
                                    if (r != run) {
                                        if (run >= 0) { atomicMin(&cmin[run], rmin); atomicMax(&cmax[run], rmax); }
                                        run = r; rmin = cxx; rmax = cxx;
                                    } else {
                                        rmin = min(rmin, cxx); rmax = max(rmax, cxx);
                                    }
                                }
                            }
                            if (run >= 0) { atomicMin(&cmin[run], rmin); atomicMax(&cmax[run], rmax); }
                            WAVE_LDS_SYNC();
                            // lane = texel row: cells of rows lane - 1 and lane touch it
                            int lo = INT_MAX, hi = INT_MIN;
                            if (lane < ncell) { lo = cmin[lane]; hi = cmax[lane]; }
                            if (lane >= 1 && lane <= ncell) { lo = min(lo, cmin[lane - 1]); hi = max(hi, cmax[lane - 1]); }
                            const int nblk = lo <= hi ? (hi - lo + 2 + 15) >> 4 : 0;   // texels lo .. hi + 1
                            const int incl = wave_scan_incl(nblk);
                            nb = __builtin_amdgcn_readlane(incl, 63);
                            fits = nb <= MAXB;
                            if (fits) {
                                const int fb = incl - nblk;
                                rowoff[lane] = 16 * fb - lo;
                                for (int i = 0; i < nblk; ++i) blk[fb + i] = ((ybase + lane) << 16) | ((lo + 16 * i) & 0xffff);
                                WAVE_LDS_SYNC();
                            }
                        }
                    }
#ifdef MFMA_REP_TABLE
                    }
#endif
                    MSTAMP(2)   // row table
                    if (!fits) {
                        MCOUNT(7, 1)   // failed trials
                        if (len == 1) { failed = true; break; }
                        len >>= 1;
                        continue;
                    }
                    MCOUNT(6, 1)    // passes
                    MCOUNT(8, nb)   // blocks
                    // The planes of this pass are done with their cells: from here on xy[j] holds their two X / Gram slots
                    // (top row | bottom row << 16; slot = x0 + rowoff of the texel row).  Computed here, not in the combine:
                    // the LDS round trip to the row table rides under the block loads of the matrix phase.
#pragma unroll
                    for (int j = 0; j < 16; ++j) {
                        if (j >= j0 && j < j1 && xy[j] != NO_CELL) {
                            const int r = cell_y(xy[j]) - ybase, cxx = cell_x(xy[j]);
                            xy[j] = (cxx + rowoff[r]) | ((cxx + rowoff[r + 1]) << 16);
                        }
                    }
                    if (MFMA_ABL_STOP == 2) { cost[h * 16] += (float)(nb + rowoff[lane & 7] + blk[lane & 7]); j0 = j1; len = min(j0 & -j0, 16 - j0); continue; }

                    // ---- X = <r, s> for the blocks of the pass, on the matrix pipe -----------------------------------
                    // (the block list rides in a register -- lane l holds entry l -- so a block's entry is a v_readlane away,
                    //  not an LDS round trip in front of every block's loads)
                    int myblk = 0;
                    auto load_block = [&](int bi, v4f(&S)[NCH > 0 ? NCH : 1], float(&T)[NTL > 0 ? NTL : 1]) {
                        const int be = __builtin_amdgcn_readlane(myblk, bi);
                        const int yy = be >> 16, xx = (int)(short)(be & 0xffff) + n;
                        const bool ok = (unsigned)xx < (unsigned)W && (unsigned)yy < (unsigned)H;
#ifdef MFMA_ABL_ROWWRAP   // timing experiment (results wrong): every texel load from 16 source rows -- an L2-resident source
                        const int t16 = ((yy & 15) * W + xx) * 16;
#else
                        const int t16 = (yy * W + xx) * 16;
#endif
                        const int vo = opaque_v(ok ? t16 + kq * HW * 16 : OOB);   // (opaque: one load with a selected offset, no branch)
#ifdef MFMA_ABL_NOLOAD   // timing experiment (results wrong): what do the texel loads of the X phase cost?
#pragma unroll
                        for (int gi = 0; gi < NCH; ++gi) S[gi] = v4f{(float)vo, 1.f, 2.f, 3.f};
#pragma unroll
                        for (int tp = 0; tp < NTL; ++tp) T[tp] = (float)vo;
                        return;
#endif
#pragma unroll
                        for (int gi = 0; gi < NCH; ++gi)
                            S[gi] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rsrc, vo, gi * 4 * HW * 16, 0));
                        const int vt = opaque_v(ok ? t16 + kq * 4 : OOB);
#pragma unroll
                        for (int tp = 0; tp < NTL; ++tp)
                            T[tp] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, vt, (4 * NCH + tp) * HW * 16, 0));
                    };
                    // one block = ceil(C/4) MFMAs in two alternating accumulator chains (a dependent f32 MFMA waits 40 cycles, an
                    // independent one issues after 32); X[texel][pixel] of the block to LDS
                    auto compute_block = [&](int bi, v4f(&S)[NCH > 0 ? NCH : 1], float(&T)[NTL > 0 ? NTL : 1], bool keep) {
                        v4f acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#ifdef MFMA_ABL_NOMFMA   // timing experiment (results wrong): what do the MFMAs of the X phase cost?
#pragma unroll
                        for (int gi = 0; gi < NCH; ++gi) { acc0 += S[gi]; }
#pragma unroll
                        for (int tp = 0; tp < NTL; ++tp) acc1[0] += T[tp];
                        if (keep) *reinterpret_cast<v4f*>(&Xs[n * XSTRIDE + 16 * bi + 4 * kq]) = acc0 + acc1;
                        return;
#endif
#pragma unroll
                        for (int gi = 0; gi < NCH; ++gi) {
                            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(S[gi][0], Rr[4 * gi + 0], acc0, 0, 0, 0);
                            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(S[gi][1], Rr[4 * gi + 1], acc1, 0, 0, 0);
                            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(S[gi][2], Rr[4 * gi + 2], acc0, 0, 0, 0);
                            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(S[gi][3], Rr[4 * gi + 3], acc1, 0, 0, 0);
                        }
#pragma unroll
                        for (int tp = 0; tp < NTL; ++tp) {
                            if (tp & 1) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(T[tp], Rr[4 * NCH + tp], acc1, 0, 0, 0);
                            else acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(T[tp], Rr[4 * NCH + tp], acc0, 0, 0, 0);
                        }
                        if (keep) *reinterpret_cast<v4f*>(&Xs[n * XSTRIDE + 16 * bi + 4 * kq]) = acc0 + acc1;
                    };
                    if (nb > 0) {
                        if (lane < 8) blk[nb + lane] = (int)0xfffe0000;   // empty blocks behind the list: loads beyond it fetch nothing
                        WAVE_LDS_SYNC();
                        myblk = blk[min(lane, MAXB + 7)];
                        WAVE_LDS_SYNC();
                        // Gram records of the pass's slots (texel = slot of a block): lane l takes slots l, l + 64, ...  The
                        // loads are issued here and land in LDS behind the multiplications.
                        // (vmcnt(0) here -- everything older has long landed: the next tile's pixel loads were issued at the top of
                        //  the tile, the previous tile's output stores before that.  With stores possibly pending the compiler
                        //  cannot count on in-order return and drains the loads of the loop below instead of overlapping them.)
                        __builtin_amdgcn_s_waitcnt(0x0F70);
                        constexpr int NGI = (MAXB * 16 + 63) / 64;
                        v4f g4[NGI];
                        float g1[NGI];
#pragma unroll
                        for (int it = 0; it < NGI; ++it) {
                            const int slot = lane + 64 * it;
                            const int be = __shfl(myblk, slot >> 4);
                            const int yy = be >> 16, xx = (int)(short)(be & 0xffff) + (slot & 15);
                            const bool ok = (unsigned)xx < (unsigned)W && (unsigned)yy < (unsigned)H && it * 64 < 16 * nb;
                            const int vg = opaque_v(ok ? (yy * W + xx) * 16 : OOB);
                            g4[it] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rsrc, vg, NPL * HW * 16, 0));
                            g1[it] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, vg, (NPL + 1) * HW * 16, 0));
                        }
                        v4f SA[NCH > 0 ? NCH : 1], SB[NCH > 0 ? NCH : 1], SC[NCH > 0 ? NCH : 1];
                        float TA[NTL > 0 ? NTL : 1], TB[NTL > 0 ? NTL : 1], TC[NTL > 0 ? NTL : 1];
#ifdef MFMA_REP_X
                        for (int rep = 0; rep < 2; ++rep) {
#endif
                        load_block(0, SA, TA);
                        load_block(1, SB, TB);
                        // two blocks in flight behind the one being multiplied.  (Unrolled over the at most MAXB blocks, with forward
                        // exits: around a loop the compiler merges the pending-load state of entry and back edge and ends up
                        // draining the prefetched blocks in front of the first multiplication of every round.)
#pragma unroll
                        for (int bi = 0; bi < MAXB; bi += 3) {
                            if (bi >= nb) break;
                            // (loads and multiplications of a round are unconditional -- the entries behind the list are empty
                            //  blocks, fetched out of range -- so that every path carries the same pending loads; only the stores
                            //  of blocks beyond the list are skipped)
                            load_block(bi + 2, SC, TC);
                            compute_block(bi, SA, TA, true);
                            load_block(bi + 3, SA, TA);
                            compute_block(bi + 1, SB, TB, bi + 1 < nb);
                            load_block(bi + 4, SB, TB);
                            compute_block(bi + 2, SC, TC, bi + 2 < nb);
                        }
#ifdef MFMA_REP_X
                        }
#endif
#pragma unroll
                        for (int it = 0; it < NGI; ++it) {
                            const int slot = lane + 64 * it;
                            if (slot < MAXB * 16) {   // (compile-time for all but the last round)
                                g4[it].w = g4[it].w + g1[it];   // (the combine only uses D1 + D2: the sum it would form itself)
                                *reinterpret_cast<v4f*>(&G4s[slot * 4]) = g4[it];
                            }
                        }
                        WAVE_LDS_SYNC();
                    }

                    MSTAMP(3)   // X on the matrix pipe
                    if (MFMA_ABL_STOP == 3) { cost[h * 16] += Xs[lane] + G4s[lane]; j0 = j1; len = min(j0 & -j0, 16 - j0); continue; }
                    // ---- combine: cost of this lane's planes of the pass ---------------------------------------------
#ifdef MFMA_REP_COMB
                    for (int rep = 0; rep < 2; ++rep) {
#endif
#pragma unroll
                    for (int j = 0; j < 16; ++j) { xy[j] = opaque_v(xy[j]); fw[j] = opaque_f(fw[j]); fn[j] = opaque_f(fn[j]); }
#pragma unroll
                    for (int j = 0; j < 16; ++j) {
                        if (j >= j0 && j < j1) {
                            // (no tap inside the image: |r|^2 -- and NaN where the position itself is not finite, as the
                            //  reference's weights inf - floor(inf) make it)
                            float q = rr + (fw[j] + fn[j]) * 0.0f;
                            if (xy[j] != NO_CELL) {
                                const int st = xy[j] & 0xffff, sb = xy[j] >> 16;
                                const float* xr = &Xs[n * XSTRIDE];
                                const float X00 = xr[st], X01 = xr[st + 1], X10 = xr[sb], X11 = xr[sb + 1];
                                const v4f G00 = *reinterpret_cast<const v4f*>(&G4s[st * 4]);
                                const float G01x = G4s[(st + 1) * 4], G01z = G4s[(st + 1) * 4 + 2];
                                const float G10x = G4s[sb * 4], G10y = G4s[sb * 4 + 1];
                                const float G11x = G4s[(sb + 1) * 4];
                                const float fe = 1.0f - fw[j], fs = 1.0f - fn[j];
                                // |sum_t w_t s_t|^2, separable in the x weights (e, w) and the y weights (s, n); fused
                                // multiply-adds (each term rounds once: at least as close to the exact value as the unfused form)
                                const float ee = fe * fe, ww = fw[j] * fw[j], ew2 = 2.0f * (fe * fw[j]);
                                const float A = __builtin_fmaf(ee, G00.x, __builtin_fmaf(ww, G01x, ew2 * G00.y));   // top row
                                const float B = __builtin_fmaf(ee, G10x, __builtin_fmaf(ww, G11x, ew2 * G10y));     // bottom row
                                const float Cq = __builtin_fmaf(ee, G00.z, __builtin_fmaf(ww, G01z, (fe * fw[j]) * G00.w));   // cross rows
                                const float Q = __builtin_fmaf(fs * fs, A, __builtin_fmaf(fn[j] * fn[j], B, (2.0f * (fs * fn[j])) * Cq));
                                const float XW = __builtin_fmaf(fs * fe, X00, __builtin_fmaf(fs * fw[j], X01,
                                                 __builtin_fmaf(fn[j] * fe, X10, (fn[j] * fw[j]) * X11)));
                                q = __builtin_fmaf(-2.0f, XW, Q) + rr;
                            }
#ifdef MFMA_REP_COMB
                            if (rep == 0) cost[h * 16 + j] = opaque_f(div_sigma(q)) * 0.0f + cost[h * 16 + j]; else
#endif
                            cost[h * 16 + j] = cost[h * 16 + j] + div_sigma(q);
                        }
                        if (j & 1) __builtin_amdgcn_sched_barrier(0);   // two planes' LDS reads in flight at a time
                    }
#ifdef MFMA_REP_COMB
                    }
#endif
                    WAVE_LDS_SYNC();   // the tables and X of this pass are dead
                    MSTAMP(4)   // combine
                    j0 = j1;
                    len = min(j0 & -j0, 16 - j0);
                }
            }
        }
        MCOUNT(9, 1)   // pixel sub-blocks
        if (MFMA_ABL_STOP == 4) { if (a.depth_out && xlive && kq == 0) { float sm = 0.f; for (int i = 0; i < NHALF * 16; ++i) sm += cost[i]; a.depth_out[(size_t)b * HW + p] = sm; } }
        else if (MFMA_ABL_STOP != 0) { if (a.depth_out && xlive && kq == 0) { float sm = 0.f; for (int i = 0; i < NHALF * 16; ++i) sm += cost[i]; a.depth_out[(size_t)b * HW + p] = sm; } }
        else if (!failed) {
#ifdef MFMA_REP_EPI
        for (int rep = 0; rep < 2; ++rep) {
        cost[0] = opaque_f(cost[0]);
#endif

        // ---- epilogue: cost store, log-softmax over D, expectation ------------------------------------------------
        // (buffer stores: one 32-bit lane offset, the plane as the scalar offset -- 64-bit per-plane pointers would be
        //  hoisted out of the row loop and spilled)
        const int ovoff = xlive ? (kq * HW + p) * 4 : OOB;
        auto plane_soff = [&](int i) { return (64 * (i / 16) + 4 * (i % 16)) * HW * 4; };   // plane 64 h + 4 j (+ kq: lane offset)
        if (a.cost_out) {
            const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc((void*)(a.cost_out + (size_t)b * D * HW), 0, D * HW * 4, 0x00020000);
#pragma unroll
            for (int i = 0; i < NHALF * 16; ++i)   // (planes beyond D lie beyond the descriptor: dropped)
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, cost[i]), rc, ovoff, plane_soff(i), 0);
        }
        if (a.logp_out || a.depth_out) {
            float mx = -INFINITY;
#pragma unroll
            for (int i = 0; i < NHALF * 16; ++i) {
                const int k = 64 * (i / 16) + 4 * (i % 16) + kq;
                if (k < D) mx = fmaxf(mx, cost[i]);
            }
            mx = fmaxf(mx, __shfl_xor(mx, 16));
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            // p_k = e_k / s with e_k = exp(c_k - max): one exp per plane; log p_k = (c_k - max) - log s
            float ssum = 0.0f, esum = 0.0f;
#pragma unroll
            for (int i = 0; i < NHALF * 16; ++i) {
                const int k = 64 * (i / 16) + 4 * (i % 16) + kq;
                const float ek = k < D ? exp_nonpos(cost[i] - mx) : 0.0f;
                ssum = ssum + ek;
                esum = __builtin_fmaf(dcl[k], ek, esum);
            }
            ssum = ssum + __shfl_xor(ssum, 16);
            ssum = ssum + __shfl_xor(ssum, 32);
            esum = esum + __shfl_xor(esum, 16);
            esum = esum + __shfl_xor(esum, 32);
            const float ls = logf(ssum);
            if (a.logp_out) {
                const __amdgpu_buffer_rsrc_t rl = __builtin_amdgcn_make_buffer_rsrc((void*)(a.logp_out + (size_t)b * D * HW), 0, D * HW * 4, 0x00020000);
#pragma unroll
                for (int i = 0; i < NHALF * 16; ++i)
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, (cost[i] - mx) - ls), rl, ovoff, plane_soff(i), 0);
            }
            if (a.depth_out && xlive && kq == 0) a.depth_out[(size_t)b * HW + p] = esum / ssum;
        }
#ifdef MFMA_REP_EPI
        }
#endif
        }
    }
    if (failed && lane == 0) {   // the gather kernel redoes the tile
        const int tiles_y = (H + 3) / 4;
        tile_flags[b * tiles_x * tiles_y + ty * tiles_x + tx] = 1;
        atomicAdd(&queue[GATHER_COUNT_SLOT], 1);
    }
    MSTAMP(5)
    if (threadIdx.x == 0) s_item[slot ^ 1] = one_each ? -1 : resolve(got_own);
    // the next tile is published, everybody is done with this one's slot.  (A raw barrier: __syncthreads() would also wait
    // for this tile's output stores to be acknowledged.)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    slot ^= 1;
    item = __builtin_amdgcn_readfirstlane(*(volatile int*)&s_item[slot]);
    if (item >= 0) load_pixel(item, Rr, ray);   // (the next tile's pixel loads fly over its setup)
    MSTAMP(10)
    }   // tiles
#ifdef MFMA_STAMPS
    if (lane == 0)
        for (int i = 0; i < 12; ++i) atomicAdd(reinterpret_cast<unsigned long long*>(queue + 16) + i, stamp_acc[i]);
#endif
#ifdef MFMA_EXIT_STAMPS   // diagnostic build: when do the persistent workgroups run out of work?  (first / last / sum of exits, count, first start)
    if (threadIdx.x == 0) {
        unsigned long long* st = reinterpret_cast<unsigned long long*>(queue + 16) + 12;
        const unsigned long long t = __builtin_amdgcn_s_memrealtime();
        atomicMax(st + 0, ~t); atomicMax(st + 1, t); atomicAdd(st + 2, t); atomicAdd(st + 3, 1ull); atomicMax(st - 1, ~exit_t0);   // (slot 11 is otherwise unused)
    }
#endif
}

template <int NPL>
hipError_t launch_npl(const SweepArgs& a, const float4* packed, int* flags, int* queue, int tiles_x, int tiles, hipStream_t stream) {
    // persistent grid: two workgroups of four waves per CU (LDS: 2 x 80 KB), a multiple of 8; fewer when there is less work
    long long nblk = ((long long)sweep_device_cus() * MFMA_WG_PER_CU + 7) & ~7ll;
    const long long need = 8ll * ((tiles + 7) / 8) * a.B;   // a workgroup per tile of the largest XCD band, times 8
    if (need <= nblk) nblk = need;
    if (a.D <= 64)
        hipLaunchKernelGGL((sweep_mfma_kernel<NPL, 1>), dim3((unsigned)nblk), dim3(256), 0, stream, a, packed, flags, queue, tiles_x, tiles);
    else
        hipLaunchKernelGGL((sweep_mfma_kernel<NPL, 2>), dim3((unsigned)nblk), dim3(256), 0, stream, a, packed, flags, queue, tiles_x, tiles);
    return hipGetLastError();
}

}  // namespace

// shapes the matrix-pipe kernel is built for: L2, D <= 128, C <= 72 (one instantiation per ceil(C/4): the reference
// features of a pixel and the texel features of a block live in registers)
constexpr int MFMA_MAX_NPL = 18;
bool sweep_mfma_supports(const SweepArgs& a) {
    const int npl = (a.C + 3) / 4;
    const long long hw = (long long)a.H * a.W;
    return a.metric == 0 && a.D <= 128 && npl <= MFMA_MAX_NPL && a.W <= 32760 && a.H <= 32760 &&
           hw * a.D * 4 < (1ll << 31) && hw * a.C * 4 < (1ll << 31) && hw * (npl + 2) * 16 < (1ll << 31) && hw * 12 < (1ll << 31);
}

namespace {
template <int NPL>
hipError_t launch_by_npl(int npl, const SweepArgs& a, const float4* packed, int* flags, int* queue, int tiles_x, int tiles, hipStream_t stream) {
    if (npl == NPL) return launch_npl<NPL>(a, packed, flags, queue, tiles_x, tiles, stream);
    if constexpr (NPL > 1) return launch_by_npl<NPL - 1>(npl, a, packed, flags, queue, tiles_x, tiles, stream);
    return hipErrorInvalidValue;
}
}  // namespace

// Launches the pre-pass (unless the workspace is already packed), the matrix-pipe kernel, then the gather kernel on the
// tiles it flagged.  Workspace layout as the tiled kernel's.
hipError_t launch_sweep_mfma(const SweepArgs& a, void* workspace, hipStream_t stream, bool packed_ready, int phases) {
    const int tiles_x = (a.W + 15) / 16, tiles_y = (a.H + 3) / 4, tiles = tiles_x * tiles_y;
    int* flags = reinterpret_cast<int*>(workspace);
    float4* packed = reinterpret_cast<float4*>(static_cast<char*>(workspace) + sweep_ws_flag_bytes(a.B, a.H, a.W));
    int* queue = reinterpret_cast<int*>(static_cast<char*>(workspace) + sweep_ws_flag_only_bytes(a.B, a.H, a.W));
    hipError_t e = hipSuccess;
    if (phases & PH_PRE) e = packed_ready ? clear_sweep_flags(a, workspace, stream) : launch_pack_c4(a, workspace, stream, /*centre=*/false);
    if (e != hipSuccess) return e;
    if (phases & PH_KERNEL) e = launch_by_npl<MFMA_MAX_NPL>((a.C + 3) / 4, a, packed, flags, queue, tiles_x, tiles, stream);
    if (e != hipSuccess || !(phases & PH_GATHER)) return e;
    SweepArgs ag = a;
    ag.packed_src = packed;
    return launch_sweep_direct_flagged(ag, flags, queue + GATHER_COUNT_SLOT, tiles_x, tiles, stream);
}

}  // namespace pdepth
