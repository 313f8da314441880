// Plane sweep in the correlation form of the L2 distance, on mean-centred features, one workgroup per block of 16 pixels:
// the default path of pdepth_sweep_{cost,dpv}_f32 for the L2 metric (C <= 72, D <= 128).
//
//   what is computed.  est_swp_volume_v4 (warping/homography.py:98-135) evaluates, per pixel p and plane k,
//           cost = sum_c ( sum_t w_t s_t[c] - r[c] )^2          (img_dis_L2_pard, :80-82; t = the four bilinear taps, :197)
//       with taps outside the image reading zero.  With mu[c] a per-channel constant, s' = s - mu, r' = r - mu and
//       Win = the weight of the taps inside the image,
//           sum_t w_t s_t - r = a - (1 - Win) mu,     a = sum_{t inside} w_t s'_t - r',
//           cost = |a|^2 - 2 (1 - Win) <a, mu> + (1 - Win)^2 |mu|^2,
//           |a|^2 = w^T G' w - 2 sum_t w_t X'_t + |r'|^2,     X'_t = <r', s'_t>,  G' = Gram terms of neighbouring s'.
//       mu = an estimate of the channel means of the source view (feature_stats_kernel, sweep_pack.hip): the terms of the
//       correlation form are then of the size of the cost itself, whatever offset the encoder's features carry (with mu = 0
//       the three terms are (mean/std)^2 times larger than their sum and cancel: tests/test_offset_features.py).  Inside the
//       image Win = 1 and the second line is the first; the correction only runs for cells on the image border
//       (<a, mu> = sum_t w_t m_t - <r', mu>, m_t = <s'_t, mu> from the pre-pass).
//
//   how.  X' for 16 neighbouring pixels x 16 consecutive texels of a source row is a 16 x 16 x C matrix product:
//       v_mfma_f32_16x16x4_f32 (exact fp32 products, fp32 accumulation).  The 64 (128) planes of the 16 pixels are spread
//       over the 256 threads of a workgroup -- thread (pixel n, tq) owns planes 4 tq .. 4 tq + 3 of each group of 64 (a
//       PASS = one source view x one group of 64 planes) -- so every phase of a pass is a few instructions long per wave,
//       and four workgroups per CU (three at D > 64) interleave:
//         positions   bit-faithful sample positions (geometry.hpp), two planes per packed instruction;
//         row table   per source row the run of texels any sample touches: LDS min / max, rows indexed modulo 64; one
//                     barrier; wave 0 cuts the runs into blocks of 16 texels while the others fetch the block's reference
//                     features from LDS; one barrier;
//         X           wave w multiplies blocks w, w + 4, ...: texel features by buffer_load_dwordx4 from the packed
//                     source (out-of-image texels: out-of-range offset = 0), half a block's loads in flight under the
//                     other half's MFMAs, the pixels' centred reference features in registers as the B operand;
//                     X[pixel][slot] to LDS; the Gram records (and <s', mu>) of the slots straight from memory to LDS
//                     (buffer_load ... lds); one barrier;
//         combine     per (pixel, plane): 4 X values, the Gram terms of the cell, the bilinear weights; the border term;
//         epilogue    cost store; log-softmax over D and E[d]: per wave partial (max, sum, sum d) of each pixel, merged
//                     across the four waves through LDS (one barrier).
//       A pass whose planes need more than MAXB blocks or 64 rows (extreme poses: epipolar segments of hundreds of texels)
//       is evaluated directly by the same workgroup behind the view loop, in the reference's own form on the centred
//       features (16-byte taps from the packed source): the kernel needs no tile flags and no second launch.
//   scheduling.  Persistent workgroups pull 16x4 tiles (four pixel blocks one after the other; single blocks on small
//       problems: CORR_SPI1_BELOW) from per-XCD queues, balanced partition as described at decode(); the last workgroup
//       to leave zeroes the queue counters, so a call on an already packed source is this one launch.
//   registers.  128 VGPRs (four waves per SIMD) is the budget everything above is written against: one straight-line pass
//       (no adaptive plane ranges, no early exits from the view loop), the per-view uniforms and the reference features
//       re-read from LDS per pass instead of held, lane-derived invariants re-derived per block (opaque_v).
#include <hip/hip_runtime.h>

#include <cstdlib>

#include <climits>

#include "../geometry.hpp"
#include "../kernels.hpp"
#include "../pack_body.hpp"
#include "../pick.hpp"
#include "sweep_corr_knobs.hpp"

namespace pdepth {

namespace {

typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));

constexpr int MAXROWS = 64;            // source rows per pass (row tables are indexed modulo 64)
constexpr int BLK_PAD = 12;            // empty entries behind the block list (loads issued beyond it fetch nothing)
constexpr int OOB = 0x7fffffff;        // buffer offset beyond every descriptor: the load returns 0
constexpr int NO_CELL = INT_MIN;
constexpr int EMPTY_BLOCK = (int)0xfffe0000;   // row -2: every texel out of range
// a plane's two X / Gram slots and what the combine needs to know about its cell, in one register
constexpr int SL_BITS = 10, SL_MASK = (1 << SL_BITS) - 1;
constexpr int SL_VALID = 1 << 20, SL_XLO = 1 << 21, SL_XHI = 1 << 22, SL_YLO = 1 << 23, SL_YHI = 1 << 24;
constexpr int SL_BORDER = SL_XLO | SL_XHI | SL_YLO | SL_YHI;

__device__ __forceinline__ int opaque_v(int x) { asm volatile("" : "+v"(x)); return x; }
__device__ __forceinline__ int opaque_s(int x) { asm volatile("" : "+s"(x)); return x; }

// every earlier LDS operation of this wave is done (LDS operations of one wave complete in order)
#define WAVE_LDS_SYNC() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
// workgroup barrier that waits for this wave's LDS traffic only: global loads and stores stay in flight
#define LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

// LDS-DMA (buffer_load ... lds): a wave-instruction moves 64 x 16 (4) bytes from memory to LDS address m0 + 16 (4) * lane, no
// registers in between.  Issued from inline asm (the compiler must not know that these loads write LDS, or it drains them
// in front of the next LDS read); counted in vmcnt like every load: the issuing wave waits for them by hand.
typedef int v4i __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void dma_b128(v4i rsrc, unsigned lds_addr, int voff, int soff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                 :: "s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff) : "memory", "m0");
}
__device__ __forceinline__ void dma_b32(v4i rsrc, unsigned lds_addr, int voff, int soff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dword %1, %2, %3 offen lds"
                 :: "s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff) : "memory", "m0");
}
__device__ __forceinline__ unsigned lds_addr_of(const void* p) {
    return (unsigned)(unsigned long long)(__attribute__((address_space(3))) const void*)p;
}

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
#define CORR_DPP_STEP(OP, ctrl) v = OP(v, __builtin_amdgcn_update_dpp(v, v, ctrl, 0xf, 0xf, false))
__device__ __forceinline__ int wave_min_i(int v) {
    CORR_DPP_STEP(min, 0xB1); CORR_DPP_STEP(min, 0x4E); CORR_DPP_STEP(min, 0x141); CORR_DPP_STEP(min, 0x140);
    return min(min(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)),
               min(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}
__device__ __forceinline__ int wave_max_i(int v) {
    CORR_DPP_STEP(max, 0xB1); CORR_DPP_STEP(max, 0x4E); CORR_DPP_STEP(max, 0x141); CORR_DPP_STEP(max, 0x140);
    return max(max(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)),
               max(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}
#undef CORR_DPP_STEP

// plane_sample_pos_fast() of geometry.hpp for two planes at a time in packed fp32 (v_pk_mul / v_pk_add / v_pk_fma_f32:
// each component rounds exactly like the scalar instruction, so the positions are bit-identical)
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

// Everything the kernel is given, in one struct that is its only argument: fields that are used once per item or pixel block
// are re-read from the kernarg segment at the point of use (KARG) instead of occupying scalar registers for the whole
// kernel, which runs out of them.
struct CorrArgs {
    SweepArgs a;
    const float4* packed;
    const float* mu_tab;
    int* queue;
    int tiles_x, ntile, spi;
    // the kernel packs the source itself (NCHW entry): pack items per batch item (0: the packed source is ready) and the
    // counters [2 b] = items of batch item b handed out, [2 b + 1] = items finished (zero at launch, zeroed on the way out)
    int npack;
    int* pack_ctr;
};
template <typename T>
__device__ __forceinline__ T cold_arg(size_t offset) {
    typedef const char __attribute__((address_space(4))) * kptr;
    typedef const volatile T __attribute__((address_space(4))) * vptr;
    return *(vptr)((kptr)__builtin_amdgcn_kernarg_segment_ptr() + offset);
}
#define KARG(type, field) cold_arg<type>(offsetof(CorrArgs, field))

constexpr int CORR_MAXV = 8;   // source views whose homography terms a workgroup keeps in LDS

// MAXB = blocks of 16 texels a pass can take; NPL = packed feature planes
template <int MAXB, int NPL>
struct __attribute__((aligned(16))) CorrLds {
    static constexpr int XSTRIDE = MAXB * 16 + 4;   // floats per pixel of the X buffer (stride / 4 odd: conflict-free b128 stores)
    static constexpr int RS_TAIL = (NPL / 4) * 256;  // reference features: [chunk g][kq][pixel][4] floats, then [tail][kq][pixel]
    float Xs[16 * XSTRIDE];      // X[pixel][slot]
    static constexpr int NSLOT = (MAXB * 16 + 63) / 64 * 64;   // (the Gram records arrive 64 slots per transfer)
    float G4s[NSLOT * 4];        // Gram record (N, H, V, D1 + D2) per slot
    float Ms[NSLOT];             // <s', mu> per slot
    float Rs[RS_TAIL + (NPL % 4) * 64 + 4];   // centred reference features of the block's 16 pixels, in B-operand order
    float rp[4 * 16 * 2];        // per wave and pixel: partial |r'|^2, <r', mu>
    float mu[80];                // channel means of the batch item in work; [72] = |mu|^2
    float xf[CORR_MAXV * 12];    // per view: K@R (9), K@t (3)
    float cst[8];                // cx, cy, 1/cx, 1/cy, W/2, H/2, sigma, 1/sigma
    float red[4 * 16 * 4];       // epilogue exchange: (max, sum, sum d) per wave and pixel
    float dcl[128];              // depth candidates
    int cmin[2][MAXROWS], cmax[2][MAXROWS];   // per cell row (modulo 64): min / max x0; two sets, alternating by pass
    int rowoff[MAXROWS + 2];     // per texel row of the pass: slot = x + rowoff
    int blk[MAXB + BLK_PAD];     // per block: (y << 16) | (x & 0xffff) of its first texel
    int ired[2][2];              // min / max cell row of the pass; two sets
    int tab[4];                  // wave 0's row table of the pass: blocks, fits, first row
    int item[2];                 // work item: current / next
    int pk;                      // pack item
    float mun[80];               // channel means of the batch item that is being packed
    unsigned char wide[64];      // per batch item: pixel blocks are 16x1 (else 8x2)
};

__device__ __forceinline__ float exp_fast(float x) { return __builtin_amdgcn_exp2f(x * 1.44269502162933349609375f); }

// NPL = packed feature planes of a source view (ceil(C / 4)); NH = groups of 64 planes (ceil(D / 64): 1 or 2), each a pass
// of its own per view: thread (n, tq) owns planes 64 h + 4 tq .. + 3 of pass h.
template <int NPL, int NH>
__global__ __launch_bounds__(256, NH == 1 ? CORR_OCC1 : CORR_OCC2) void sweep_corr_kernel(CorrArgs ca) {
    constexpr int NCH = NPL / 4, NTL = NPL % 4;   // chunks of 16 channels (4 MFMAs per 16-byte load), left-over planes of 4
    constexpr int LA = (NCH + 1) / 2, LB = NCH - LA;   // a block's loads in two halves: chunks [0, LA) | chunks [LA, NCH) + left-overs
    constexpr int MCH = (4 * NPL + 15) / 16;      // channels per thread of the cooperative reference load
    constexpr int MAXB = NH == 1 ? CORR_MAXB1 : CORR_MAXB2;
    constexpr int BPW = (MAXB + 3) / 4;           // blocks per wave and pass
    constexpr int NC = 4 * NH;                    // planes (costs) per thread
    typedef CorrLds<MAXB, NPL> Lds;
    constexpr int XSTRIDE = Lds::XSTRIDE;
    static_assert(MAXB * 16 + 16 <= (1 << SL_BITS), "slot bits");
    __shared__ Lds L;
    if (poison_on_foreign_layout(ca.a, ca.queue, LAYOUT_C4_CENTRED)) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int D = KARG(int, a.D), H = KARG(int, a.H), W = KARG(int, a.W), V = KARG(int, a.V), C = KARG(int, a.C);

    {
        const float* dc = KARG(const float*, a.d_candi);
        for (int k = tid; k < 64 * NH; k += 256) L.dcl[k] = dc[min(k, D - 1)];
    }
    if (tid < 64) {
        L.cmin[0][tid] = INT_MAX; L.cmax[0][tid] = INT_MIN;
        L.cmin[1][tid] = INT_MAX; L.cmax[1][tid] = INT_MIN;
        L.wide[tid] = 1;
    }
    if (tid < 2) { L.ired[tid][0] = INT_MAX; L.ired[tid][1] = INT_MIN; }
    __syncthreads();
    // Shape of the pixel blocks of a batch item: 16x1 where the epipolar lines of view 0 run along the source rows (a
    // rectified pair: the 16 pixels of a row share two source rows), else 8x2 (pick.hpp; any choice is correct).
    if ((int)(tid >> 2) < min(KARG(int, a.B), 64) && epipolar_probe_is_steep(ca.a, tid >> 2, tid & 3)) L.wide[tid >> 2] = 0;

    // ---- work queue (per XCD, as the other persistent kernels of this library) ----------------------------------------
    // Workgroups are dealt round-robin over the 8 XCDs; XCD q owns the tiles of band q (its own queue counter), so that
    // neighbouring tiles, whose source texels overlap, meet in that XCD's L2.  A workgroup whose queue is exhausted takes
    // items of the others.  m = items per tile (1: a tile's four pixel blocks in sequence; 4: one block per item).
    const int xcd = blockIdx.x & 7;
    auto band_tiles_of = [&](int q) { const int nt = KARG(int, ntile); return (nt >> 3) + (q < (nt & 7) ? 1 : 0); };
    auto band_first_of = [&](int q) {
        const int nt = KARG(int, ntile), qq = nt >> 3, rr8 = nt & 7;
        return q < rr8 ? q * (qq + 1) : rr8 * (qq + 1) + (q - rr8) * qq;
    };
    bool own_done = false;
    auto steal = [&]() -> int {   // (queue << 28) | index in the queue, or -1: every queue is exhausted
        int* queue = KARG(int*, queue);
        const int mB = (4 / ca.spi) * KARG(int, a.B);
        for (int j = 1; j < 8; ++j) {
            const int q = (xcd + j) & 7, nq = band_tiles_of(q) * mB;
            if (*(volatile int*)&queue[q] >= nq) continue;
            const int got = atomicAdd(&queue[q], 1);
            if (got < nq) return (q << 28) | got;
        }
        return -1;
    };
    auto resolve = [&](int got) -> int {
        const int n_own = band_tiles_of(xcd) * (4 / ca.spi) * KARG(int, a.B);
        if (!own_done && got < n_own) return (xcd << 28) | got;
        own_done = true;
        return steal();
    };
    // floor(nn / dd) for 0 <= nn < 2^22, 0 < dd (an integer divide costs ~40 dependent instructions)
    auto fdiv = [](int nn, int dd) { return (int)(((float)nn + 0.5f) * __builtin_amdgcn_rcpf((float)dd)); };
    auto decode = [&](int item, int& b_, int& tx_, int& ty_, int& sub0_) {
        const int ntile = KARG(int, ntile), tiles_x = KARG(int, tiles_x), spi = ca.spi;
        const int m = 4 / spi, msh = spi == 1 ? 2 : 0, qq = ntile >> 3, rr8 = ntile & 7;
        const bool small_idx = (long long)ntile * m * KARG(int, a.B) < (1ll << 22);
        const int tiles_y_ = (H + 3) / 4;
        const int q_ = item >> 28, iq = item & 0x0fffffff, band_tiles = band_tiles_of(q_), per_b = band_tiles * m;
        b_ = small_idx ? fdiv(iq, per_b) : iq / per_b;
        const int rem = iq - b_ * per_b, ti = rem >> msh;
        sub0_ = (rem & (m - 1)) * spi;
        int tile = band_first_of(q_) + ti;
        if (CORR_HALF_BANDS && rr8 == 0 && tiles_y_ % 16 == 0 && tiles_y_ * tiles_x == ntile) {
            // XCD q owns half-bands q and 8 + q of the image's 16: on a forward motion the cost of a tile grows with its
            // distance from the image centre, and this way every XCD gets the same mix; the heavier half first and, inside
            // a half, columns from both image borders inwards.  (Any static partition is valid: dry queues steal.)
            const int hb_rows = tiles_y_ / 16, half_tiles = hb_rows * tiles_x;
            const int second = ti >= half_tiles ? 1 : 0, tih = ti - second * half_tiles;
            const int hbi = (q_ < 4) == (second == 0) ? q_ : 8 + q_;
            const int cc = small_idx ? fdiv(tih, hb_rows) : tih / hb_rows, r_ = tih - cc * hb_rows;
            const int col = (cc & 1) ? tiles_x - 1 - (cc >> 1) : (cc >> 1);
            tile = (hbi * hb_rows + r_) * tiles_x + col;
        } else if (rr8 == 0 && qq % tiles_x == 0) {   // the band is a whole number of tile rows: column by column
            const int band_rows = qq / tiles_x, tc = small_idx ? fdiv(ti, band_rows) : ti / band_rows;
            tile = (q_ * band_rows + (ti - tc * band_rows)) * tiles_x + tc;
        }
        ty_ = small_idx ? fdiv(tile, tiles_x) : tile / tiles_x;
        tx_ = tile - ty_ * tiles_x;
    };
    // (when the grid covers every item -- small problems -- workgroup i takes item i of its XCD's band: no atomics)
    const bool one_each = (long long)gridDim.x >= 8ll * ((KARG(int, ntile) + 7) / 8) * (4 / ca.spi) * KARG(int, a.B);
    __syncthreads();
    // ---- the pre-pass inside the sweep (NCHW entry) -------------------------------------------------------------------
    // The source of batch item b + 1 is packed while item b is swept: a workgroup that finishes a pixel block of item b
    // takes a pack item of b + 1 (256 texels of one view: sweep_pack.hip's layout, pack_body.hpp's arithmetic) as long as
    // there are any, and a workgroup that reaches item b for the first time takes what is left of b's own and then waits
    // for the count of finished items.  No item waits on anything once it is taken, so the wait ends.  Visibility across the XCDs' L2s:
    // the packed planes are stored write-through (sc1) and released at agent scope before the count goes up; no line of
    // them is in any cache before that -- they are first read after the wait (caches are clean at launch).
    const int npack = KARG(int, npack);
    int mun_b = -1;       // batch item whose means are in L.mun
    int pack_dry = -1;    // the pack queues of batch items <= this are exhausted
    auto pack_take = [&](int pb) -> bool {
        if (tid == 0) L.pk = atomicAdd(&KARG(int*, pack_ctr)[2 * pb], 1);
        LDS_BARRIER();
        const int ip = __builtin_amdgcn_readfirstlane(*(volatile int*)&L.pk);
        if (ip >= npack) { pack_dry = pb; return false; }
        if (mun_b != pb) {
            mun_b = pb;
            if (tid < 80) L.mun[tid] = ca.mu_tab[pb * STATS_STRIDE + tid];
            LDS_BARRIER();
        }
        const int HW = H * W, nblk = (HW + 255) >> 8;
        const int vsel = ip / nblk, pix = (ip - vsel * nblk) * 256 + tid;
        if (pix < HW) {
            const int y = pix / W, x = pix - y * W;
            const float* s = KARG(const float*, a.src) + (size_t)pb * KARG(long long, a.src_bstride) + (size_t)vsel * KARG(long long, a.src_vstride) + pix;
            const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(
                (void*)(ca.packed + ((size_t)pb * V + vsel) * (NPL + 2) * HW), 0, (NPL + 2) * HW * 16, 0x00020000);
            pack_texel<true>(s, C, HW, W, x + 1 < W, y + 1 < H, L.mun, [&](int g, float4 q) {
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4i, q), ro, pix * 16, g * HW * 16, CORR_PACK_AUX);
            });
        }
        // (a release fence at agent scope here writes the whole L2 back -- the sweep's own output included -- per item:
        //  measured 0.96 ms per call instead of 0.67)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's write-through stores have reached memory
        LDS_BARRIER();
        if (tid == 0) atomicAdd(&KARG(int*, pack_ctr)[2 * pb + 1], 1);
        return true;
    };

#ifdef CORR_STAMPS
    unsigned long long stamp_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long stamp_t = __builtin_amdgcn_s_memrealtime();
#endif
    int slot_par = 0;
    int pt = 0;           // running pass counter: selects the set of row-table arrays
    int n_direct = 0;     // (thread 0) pixel blocks evaluated directly
    int b_tables = -1;    // batch item whose tables (means, homography terms, camera constants) are in LDS
    // (thread 0) the atomic on the own queue is issued at the top of an item and its result looked at when the item is done
    int got_own = (tid == 0 && !one_each) ? atomicAdd(&KARG(int*, queue)[xcd], 1) : (int)(blockIdx.x >> 3);
    bool first = true;

    for (;;) {
        if (tid == 0) {
            const int n_own = band_tiles_of(xcd) * (4 / ca.spi) * KARG(int, a.B);
            L.item[slot_par] = one_each ? (first && got_own < n_own ? (xcd << 28) | got_own : -1) : resolve(got_own);
        }
        first = false;
        LDS_BARRIER();   // the item is published; every wave is done with the previous one's LDS
        const int item = __builtin_amdgcn_readfirstlane(*(volatile int*)&L.item[slot_par]);
        slot_par ^= 1;
        if (item < 0) break;
        if (tid == 0 && !one_each && !own_done) got_own = atomicAdd(&KARG(int*, queue)[xcd], 1);
        CSTAMP(0)   // queue: publish + barrier
        int b, tx, ty, sub0;
        decode(item, b, tx, ty, sub0);
        // per BATCH item, for every wave: channel means, the views' homography terms, the camera constants (visible behind the
        // barrier in front of the first block's centring).  Items come out of the queues batch item by batch item: the tables
        // are rebuilt four times per launch, not once per tile.
        const bool new_b = b != b_tables;
        if (npack > 0) {
            // the first item of batch item b: what is left of b's pack items, then the wait for all of them; every other
            // item: one pack item of b + 1, if there is any left
            const int pb = new_b ? b : b + 1;
            if (pb < KARG(int, a.B))
                while (pack_dry < pb && pack_take(pb) && new_b) {}
            if (new_b) {
                if (tid == 0) {
                    int* done = &KARG(int*, pack_ctr)[2 * b + 1];
                    // (bounded: a call that got here with broken counters ends with wrong numbers and a raised slot, not with a hung GPU)
                    int spins = 0;
                    while (atomicAdd(done, 0) < npack && ++spins < (1 << 20)) __builtin_amdgcn_s_sleep(8);
                    if (spins >= (1 << 20)) atomicAdd(&KARG(int*, queue)[CORR_PACK_TIMEOUT_SLOT], 1);
                }
                LDS_BARRIER();
            }
        }
        if (new_b) {
            b_tables = b;
            if (tid < 72) L.mu[tid] = ca.mu_tab[b * STATS_STRIDE + tid];
            else if (tid < 80 && tid != 72) L.mu[tid] = 0.0f;
            if (wave == 1) {   // |mu|^2
                const float* mt = ca.mu_tab + b * STATS_STRIDE;
                float m2 = 0.0f;
                if (lane < 36) { const float u0 = mt[lane], u1 = mt[lane + 36]; m2 = __builtin_fmaf(u0, u0, u1 * u1); }
#pragma unroll
                for (int sh = 32; sh >= 1; sh >>= 1) m2 = m2 + __shfl_xor(m2, sh);
                if (lane == 0) L.mu[72] = m2;
            }
            if (tid >= 128 && tid < 128 + V) {
                const int v = tid - 128;
                ViewXform xf;
                make_view_xform(KARG(const float*, a.K) + b * 9, KARG(const float*, a.R) + ((size_t)b * V + v) * 9,
                                KARG(const float*, a.t) + ((size_t)b * V + v) * 3, ca.a.blas_mode, xf);
#pragma unroll
                for (int i = 0; i < 9; ++i) L.xf[v * 12 + i] = xf.kr[i];
#pragma unroll
                for (int i = 0; i < 3; ++i) L.xf[v * 12 + 9 + i] = xf.kt[i];
            }
            if (tid == 255) {
                const float* const cxcy_ = KARG(const float*, a.cxcy);
                const float cx = cxcy_[b * 2 + 0], cy = cxcy_[b * 2 + 1], sg = KARG(float, a.sigma);
                L.cst[0] = cx; L.cst[1] = cy; L.cst[2] = refined_rcp(cx); L.cst[3] = refined_rcp(cy);
                L.cst[4] = (float)W / 2.0f; L.cst[5] = (float)H / 2.0f; L.cst[6] = sg; L.cst[7] = refined_rcp(sg);
            }
        }
        const bool wide = b < 64 ? L.wide[b] != 0 : false;
        bool item_ready = !new_b;
        float ray[3], rv[MCH];
        const int spi = ca.spi;

        for (int sub = sub0; sub < sub0 + spi; ++sub) {
            if ((wide ? ty * 4 + sub : ty * 4 + 2 * (sub >> 1)) >= H) continue;   // the block lies below the image (uniform)
            // lane roles: in the vector phases thread (n, tq) owns pixel n of the block and planes 64 h + 4 tq .. + 3; in the
            // matrix phase lane (n, kq) of a wave feeds texel / pixel n and channel slice kq.  (opaque: the optimiser
            // otherwise hoists every lane-derived invariant of the phases -- masks, LDS addresses, offsets -- to the top of
            // the kernel and spills them)
            const int tid = opaque_v((int)threadIdx.x), lane = tid & 63;   // (shadow the kernel's: re-derived per pixel block)
            const int n = lane & 15, kq = lane >> 4, tq = wave * 4 + kq;
            const int HW = opaque_s(H * W);
            const int x = wide ? tx * 16 + n : tx * 16 + 8 * (sub & 1) + (n & 7);
            const int y = wide ? ty * 4 + sub : ty * 4 + 2 * (sub >> 1) + (n >> 3);
            const bool xlive = x < W && y < H;
            const int p = min(y, H - 1) * W + min(x, W - 1);
            // the pixel's ray, and this thread's share of the block's reference features: channels tq, tq + 16, ... of pixel n
            // (buffer loads: 32-bit offsets, the channel as the scalar offset; channels beyond C = 0).  (Issuing them for the next
            // block under this block's epilogue was measured: slower, the eight registers they hold spill.)
            auto issue_pixel_loads = [&](int p_) {
                const __amdgpu_buffer_rsrc_t rray =
                    __builtin_amdgcn_make_buffer_rsrc((void*)(ca.a.rays + (size_t)b * 3 * HW), 0, 3 * HW * 4, 0x00020000);
#pragma unroll
                for (int i = 0; i < 3; ++i) ray[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rray, p_ * 4, i * HW * 4, 0));
                const __amdgpu_buffer_rsrc_t rref =
                    __builtin_amdgcn_make_buffer_rsrc((void*)(ca.a.ref + (size_t)b * ca.a.ref_bstride), 0,
                                                      C * HW * 4, 0x00020000);
#pragma unroll
                for (int mm = 0; mm < MCH; ++mm)
                    rv[mm] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                        rref, tq + 16 * mm < C ? (tq * HW + p_) * 4 : OOB, 16 * mm * HW * 4, 0));
            };
            issue_pixel_loads(p);
            CSTAMP(1)   // item set-up, pixel loads issued
            float rr = 0.0f, rho = 0.0f;   // |r'|^2 and <r', mu> of the pixel (set with the first pass)
            bool centred = false;
            unsigned failmask = 0;         // uniform over the workgroup: passes (view v, plane group h) that did not fit the row tables

            float cost[NC];
#pragma unroll
            for (int j = 0; j < NC; ++j) cost[j] = 0.0f;

            for (int v = 0; v < V; ++v) {
                const float4* srcv = ca.packed + ((size_t)b * V + v) * (NPL + 2) * HW;
                const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)srcv, 0, (NPL + 2) * HW * 16, 0x00020000);

#pragma unroll
                for (int h = 0; h < NH; ++h) {
                    const int par = pt & 1;
                    if (!item_ready) { LDS_BARRIER(); item_ready = true; }   // (the item's tables above are in LDS)
                    // ---- sample positions of this thread's planes of the pass (NO_CELL: no tap in the image, plane beyond D,
                    //      pixel beyond the image)
                    int cell[4];
                    float fw[4], fn[4];
                    {
                        ViewXform xf;
                        float t2a, t2b, t2c;
                        {
                            const v4f k0 = *reinterpret_cast<const v4f*>(&L.xf[v * 12]), k1 = *reinterpret_cast<const v4f*>(&L.xf[v * 12 + 4]),
                                      k2 = *reinterpret_cast<const v4f*>(&L.xf[v * 12 + 8]);
                            xf.kr[0] = k0.x; xf.kr[1] = k0.y; xf.kr[2] = k0.z; xf.kr[3] = k0.w; xf.kr[4] = k1.x; xf.kr[5] = k1.y;
                            xf.kr[6] = k1.z; xf.kr[7] = k1.w; xf.kr[8] = k2.x; xf.kt[0] = k2.y; xf.kt[1] = k2.z; xf.kt[2] = k2.w;
                            xf.separate = ca.a.blas_mode;
                            ray_term2(xf, ray[0], ray[1], ray[2], t2a, t2b, t2c);
                        }
                        const v4f c0 = *reinterpret_cast<const v4f*>(&L.cst[0]);
                        const v2f c1 = *reinterpret_cast<const v2f*>(&L.cst[4]);
#pragma unroll
                        for (int j = 0; j < 4; j += 2) {
                            const int k = 64 * h + 4 * tq + j;
                            v2f ix, iy;
                            plane_sample_pos_fast2(xf, t2a, t2b, t2c, v2f{L.dcl[k], L.dcl[k + 1]}, c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, ix, iy);
                            cell[j] = cell_of(ix.x, iy.x, W, H, fw[j], fn[j]);
                            cell[j + 1] = cell_of(ix.y, iy.y, W, H, fw[j + 1], fn[j + 1]);
                            if (k >= D || !xlive) cell[j] = NO_CELL;
                            if (k + 1 >= D || !xlive) cell[j + 1] = NO_CELL;
                        }
                        // (pinned: the optimiser otherwise carries the positions AND their floors to the combine instead of the fractions)
#pragma unroll
                        for (int j = 0; j < 4; ++j) { asm volatile("" : "+v"(fw[j])); asm volatile("" : "+v"(fn[j])); }
                    }
                    CSTAMP(2)   // (wait for the ray) sample positions
                    // ---- row table: contributions of this thread's planes --------------------------------------------------
                    {
                        int lmin = INT_MAX, lmax = INT_MIN;
                        int run = INT_MIN, rmin = 0, rmax = 0;   // consecutive planes of equal row are folded first
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            if (cell[j] != NO_CELL) {
                                const int cyy = cell_y(cell[j]), cxx = cell_x(cell[j]);
                                lmin = min(lmin, cyy); lmax = max(lmax, cyy);
                                if (cyy != run) {
                                    if (run != INT_MIN) { atomicMin(&L.cmin[par][run & 63], rmin); atomicMax(&L.cmax[par][run & 63], rmax); }
                                    run = cyy; rmin = cxx; rmax = cxx;
                                } else {
                                    rmin = min(rmin, cxx); rmax = max(rmax, cxx);
                                }
                            }
                        }
                        if (run != INT_MIN) { atomicMin(&L.cmin[par][run & 63], rmin); atomicMax(&L.cmax[par][run & 63], rmax); }
                        const int wmin = wave_min_i(lmin), wmax = wave_max_i(lmax);
                        if (lane == 0 && wmin <= wmax) { atomicMin(&L.ired[par][0], wmin); atomicMax(&L.ired[par][1], wmax); }
                    }
                    if (!centred) {
                        // (first pass of the block) this thread's channels of pixel n, centred, into the B-operand image of the
                        // block: channel c = 16 g + 4 kq' + i belongs to lane (n, kq') of every wave, component i of chunk g --
                        // for c = tq + 16 mm that is g = mm, kq' = wave, i = kq; and its share of |r'|^2 and <r', mu>
                        float pr = 0.0f, ph = 0.0f;
#pragma unroll
                        for (int mm = 0; mm < MCH; ++mm) {
                            // (the last round only covers the left-over planes; channels beyond C: 0 - 0)
                            const bool has = mm < NCH || tq < 4 * NTL;
                            const float u = has ? L.mu[min(tq + 16 * mm, 71)] : 0.0f;
                            const float r = has ? rv[mm] - u : 0.0f;
                            pr = __builtin_fmaf(r, r, pr);
                            ph = __builtin_fmaf(r, u, ph);
                            if (mm < NCH) L.Rs[((mm * 4 + wave) * 16 + n) * 4 + kq] = r;
                            else if (wave < NTL) L.Rs[Lds::RS_TAIL + (wave * 4 + kq) * 16 + n] = r;
                        }
                        pr = pr + __shfl_xor(pr, 16); pr = pr + __shfl_xor(pr, 32);
                        ph = ph + __shfl_xor(ph, 16); ph = ph + __shfl_xor(ph, 32);
                        if (kq == 0) *reinterpret_cast<v2f*>(&L.rp[(wave * 16 + n) * 2]) = v2f{pr, ph};
                    }
                    CSTAMP(3)   // table atomics, (wait for the reference features) centring
                    LDS_BARRIER();
                    CSTAMP(4)   // barrier: tables complete
                    // ---- the row table is cut into blocks by wave 0 alone (the same ~250 instructions in every wave were 15 % of the
                    //      kernel's vector work); the other waves fetch the block's reference features meanwhile
                    if (wave == 0) {
                        const int yb = __builtin_amdgcn_readfirstlane(L.ired[par][0]), yt = __builtin_amdgcn_readfirstlane(L.ired[par][1]);
                        int nb_ = 0;
                        bool fits_ = true, border_ = false;   // border_: a cell of the pass lies on the image border (the <s', mu> records are needed)
                        if (yb <= yt) {
                            const int ncell = yt - yb + 1;
                            if (ncell + 1 > MAXB) {   // (every texel row takes a block; also keeps the modulo-64 rows apart)
                                fits_ = false;
                            } else {
                                // lane = texel row yb + lane: the cells of rows lane - 1 and lane touch it
                                int lo = INT_MAX, hi = INT_MIN;
                                if (lane < ncell) { lo = L.cmin[par][(yb + lane) & 63]; hi = L.cmax[par][(yb + lane) & 63]; }
                                if (lane >= 1 && lane <= ncell) {
                                    lo = min(lo, L.cmin[par][(yb + lane - 1) & 63]);
                                    hi = max(hi, L.cmax[par][(yb + lane - 1) & 63]);
                                }
                                const int nblk = lo <= hi ? (hi - lo + 2 + 15) >> 4 : 0;   // texels lo .. hi + 1
                                border_ = yb < 0 || yt >= H - 1 || __builtin_amdgcn_ballot_w64(lo <= hi && (lo < 0 || hi >= W - 1)) != 0;
                                const int incl = wave_scan_incl(nblk);
                                nb_ = __builtin_amdgcn_readlane(incl, 63);
                                fits_ = nb_ <= MAXB;
                                if (fits_) {
                                    const int fb = incl - nblk;
                                    if (lane <= ncell) L.rowoff[lane] = 16 * fb - lo;
#pragma unroll
                                    for (int i = 0; i < 3; ++i)   // (a row rarely needs more than three blocks: no loop for those)
                                        if (i < nblk) L.blk[fb + i] = ((yb + lane) << 16) | ((lo + 16 * i) & 0xffff);
                                    for (int i = 3; i < nblk; ++i) L.blk[fb + i] = ((yb + lane) << 16) | ((lo + 16 * i) & 0xffff);
                                    if (lane < BLK_PAD) L.blk[nb_ + lane] = EMPTY_BLOCK;
                                }
                            }
                        }
                        if (lane == 0) { L.tab[0] = nb_; L.tab[1] = fits_ ? 1 : 0; L.tab[2] = yb; L.tab[3] = border_ ? 1 : 0; }
                        // (only this wave reads the min / max tables: they are dead now)
                        L.cmin[par][lane] = INT_MAX; L.cmax[par][lane] = INT_MIN;
                        if (lane == 0) { L.ired[par][0] = INT_MAX; L.ired[par][1] = INT_MIN; }
                    }
                    // the block's reference features from LDS, as lane (n, kq) feeds them to the matrix pipe (every pass anew: they
                    // would otherwise occupy registers through the vector phases of every further view)
                    float Rr[NPL];
#pragma unroll
                    for (int g = 0; g < NCH; ++g) {
                        const v4f r4 = *reinterpret_cast<const v4f*>(&L.Rs[((g * 4 + kq) * 16 + n) * 4]);
                        Rr[4 * g + 0] = r4.x; Rr[4 * g + 1] = r4.y; Rr[4 * g + 2] = r4.z; Rr[4 * g + 3] = r4.w;
                    }
#pragma unroll
                    for (int tp = 0; tp < NTL; ++tp) Rr[4 * NCH + tp] = L.Rs[Lds::RS_TAIL + (tp * 4 + kq) * 16 + n];
                    if (!centred) {
#pragma unroll
                        for (int w = 0; w < 4; ++w) {
                            const v2f pp = *reinterpret_cast<const v2f*>(&L.rp[(w * 16 + n) * 2]);
                            rr = rr + pp.x; rho = rho + pp.y;
                        }
                        centred = true;
                    }
                    LDS_BARRIER();   // the block list is complete
                    const int nb = __builtin_amdgcn_readfirstlane(L.tab[0]), ybase = __builtin_amdgcn_readfirstlane(L.tab[2]);
                    bool fits = __builtin_amdgcn_readfirstlane(L.tab[1]) != 0;
#ifdef CORR_FORCE_DIRECT   // test build: every pass takes the direct evaluation
                    fits = false;
#endif
                    CSTAMP(5)   // scan
                    // ---- X' = <r', s'> for the blocks of the pass, on the matrix pipe ---------------------------------------
                    const bool go = fits && nb > 0;
                    const bool border = __builtin_amdgcn_readfirstlane(L.tab[3]) != 0;
                    int sl[4];
                    {
                        const int myblk = L.blk[min(lane, MAXB + BLK_PAD - 1)];   // the block list in a register: entry l in lane l
                        // Gram records (N, H, V, D1 + D2) and <s', mu> of the pass's slots (texel = slot of a block), straight
                        // from the packed source into LDS: wave w moves slots 64 c .. 64 c + 63 for c = w, w + 4, ...
                        {
                            v4i rs4;   // the descriptor of rsrc, spelled out for the inline asm
                            {
                                const unsigned long long pa = reinterpret_cast<unsigned long long>(srcv);
                                rs4.x = (int)(unsigned)pa; rs4.y = (int)(unsigned)(pa >> 32) & 0xffff; rs4.z = (NPL + 2) * HW * 16; rs4.w = 0x00020000;
                            }
#pragma unroll
                            for (int c = 0; c < (MAXB * 16 + 255) / 256; ++c) {
                                const int c64 = (wave + 4 * c) * 64;
                                if (go && c64 < 16 * nb) {   // (uniform per wave)
                                    const int slot = c64 + lane;
                                    const int be = L.blk[min(slot >> 4, MAXB + BLK_PAD - 1)];
                                    const int yy = be >> 16, xx = (int)(short)(be & 0xffff) + (slot & 15);
                                    const bool ok = (unsigned)xx < (unsigned)W && (unsigned)yy < (unsigned)H && slot < 16 * nb;
                                    const int vg = ok ? (yy * W + xx) * 16 : OOB;
                                    dma_b128(rs4, lds_addr_of(&L.G4s[c64 * 4]), vg, NPL * HW * 16);
                                    if (CORR_MS_ALWAYS || border) dma_b32(rs4, lds_addr_of(&L.Ms[c64]), vg, (NPL + 1) * HW * 16);
                                }
                            }
                        }
                        // a block's texel features in two halves (chunks [0, LA) | chunks [LA, NCH) + left-overs): while one half is
                        // multiplied the other half's loads are in flight -- half the staging registers of whole blocks
                        v4f SA[LA > 0 ? LA : 1], SB[LB > 0 ? LB : 1];
                        float TB[NTL > 0 ? NTL : 1];
                        int vo = OOB, vt = OOB;   // lane offsets of the block whose loads are being issued
                        auto prep = [&](int bi) {
                            const int be = __builtin_amdgcn_readlane(myblk, bi);
                            const int yy = be >> 16, xx = (int)(short)(be & 0xffff) + n;
                            const bool ok = (unsigned)xx < (unsigned)W && (unsigned)yy < (unsigned)H;
                            const int t16 = (yy * W + xx) * 16;
                            vo = opaque_v(ok ? t16 + kq * HW * 16 : OOB);   // (opaque: one load with a selected offset, no branch)
                            if (NTL > 0) vt = opaque_v(ok ? t16 + kq * 4 : OOB);
                        };
                        auto load_a = [&]() {
#pragma unroll
                            for (int gi = 0; gi < LA; ++gi)
                                SA[gi] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rsrc, vo, gi * 4 * HW * 16, 0));
                        };
                        auto load_b = [&]() {
#pragma unroll
                            for (int gi = 0; gi < LB; ++gi)
                                SB[gi] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rsrc, vo, (LA + gi) * 4 * HW * 16, 0));
#pragma unroll
                            for (int tp = 0; tp < NTL; ++tp)
                                TB[tp] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, vt, (4 * NCH + tp) * HW * 16, 0));
                        };
                        // ceil(C/4) MFMAs per block in two alternating accumulator chains (a dependent f32 MFMA waits 40 cycles, an
                        // independent one issues after 32)
                        v4f acc0, acc1;
                        auto mul_a = [&]() {
#pragma unroll
                            for (int gi = 0; gi < LA; ++gi) {
                                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(SA[gi][0], Rr[4 * gi + 0], acc0, 0, 0, 0);
                                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(SA[gi][1], Rr[4 * gi + 1], acc1, 0, 0, 0);
                                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(SA[gi][2], Rr[4 * gi + 2], acc0, 0, 0, 0);
                                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(SA[gi][3], Rr[4 * gi + 3], acc1, 0, 0, 0);
                            }
                        };
                        auto mul_b = [&]() {
#pragma unroll
                            for (int gi = 0; gi < LB; ++gi) {
                                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(SB[gi][0], Rr[4 * (LA + gi) + 0], acc0, 0, 0, 0);
                                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(SB[gi][1], Rr[4 * (LA + gi) + 1], acc1, 0, 0, 0);
                                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(SB[gi][2], Rr[4 * (LA + gi) + 2], acc0, 0, 0, 0);
                                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(SB[gi][3], Rr[4 * (LA + gi) + 3], acc1, 0, 0, 0);
                            }
#pragma unroll
                            for (int tp = 0; tp < NTL; ++tp) {
                                if (tp & 1) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(TB[tp], Rr[4 * NCH + tp], acc1, 0, 0, 0);
                                else acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(TB[tp], Rr[4 * NCH + tp], acc0, 0, 0, 0);
                            }
                        };
                        // the first block's loads and the Gram records are in flight while the slots are looked up
                        // (a wave in the matrix phase issues ahead of waves in the vector phases: its loads are the longest wait of a pass)
                        if (CORR_XPRIO) __builtin_amdgcn_s_setprio(CORR_XPRIO);
                        if (go) { prep(wave); load_a(); load_b(); }
                        // the two X / Gram slots of this thread's planes and the border flags of their cells, one register each
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            // (branch-free: a plane without a cell looks up row 0 and drops the result)
                            const bool has = fits && cell[j] != NO_CELL;
                            const int cyy = cell_y(cell[j]), cxx = cell_x(cell[j]), r = has ? cyy - ybase : 0;
                            const int v = (cxx + L.rowoff[r]) | ((cxx + L.rowoff[r + 1]) << SL_BITS) | SL_VALID | (cxx < 0 ? SL_XLO : 0) |
                                          (cxx >= W - 1 ? SL_XHI : 0) | (cyy < 0 ? SL_YLO : 0) | (cyy >= H - 1 ? SL_YHI : 0);
                            sl[j] = has ? v : 0;
                        }
                        if (go) {
                        // wave w: blocks w, w + 4, ... (unrolled with forward exits; loads beyond the list hit the empty entries behind
                        // it and fetch nothing)
#pragma unroll
                        for (int i = 0; i < BPW; ++i) {
                            const int bi = wave + 4 * i;
                            if (bi >= nb) break;
                            acc0 = v4f{0.f, 0.f, 0.f, 0.f}; acc1 = v4f{0.f, 0.f, 0.f, 0.f};
                            if (i + 1 < BPW) prep(bi + 4);
                            mul_a();
                            if (i + 1 < BPW) load_a();
                            mul_b();
                            if (i + 1 < BPW) load_b();
                            *reinterpret_cast<v4f*>(&L.Xs[n * XSTRIDE + 16 * bi + 4 * kq]) = acc0 + acc1;   // X[texel][pixel] of the block
                        }
                        CSTAMP(6)   // X: loads + multiplications
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's Gram records have landed in LDS
                        }
                    }
                    if (CORR_XPRIO) __builtin_amdgcn_s_setprio(0);
                    CSTAMP(7)   // wait for the Gram records
                    LDS_BARRIER();   // X and the Gram records of the pass are complete (or: every wave has seen that it does not fit)
                    ++pt;
                    CSTAMP(8)   // barrier: X complete

                    // ---- combine: cost of this thread's planes of the pass ------------------------------------------------
                    if (!fits) {
                        failmask |= 1u << (v * NH + h);   // (evaluated directly behind the view loop: direct_pass below)
                    } else {
                        const v2f sg = *reinterpret_cast<const v2f*>(&L.cst[6]);   // sigma, 1 / sigma
                        const float M2 = L.mu[72];
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            // (no tap inside the image: the taps read zero, cost = |r|^2 = |r' + mu|^2 -- and NaN where the
                            //  position itself is not finite, as the reference's weights inf - floor(inf) make it)
                            float q = (__builtin_fmaf(2.0f, rho, rr) + M2) + (fw[j] + fn[j]) * 0.0f;
                            if (sl[j] & SL_VALID) {
                                const int st = sl[j] & SL_MASK, sb = (sl[j] >> SL_BITS) & SL_MASK;
                                const float* xr = &L.Xs[n * XSTRIDE];
                                const float X00 = xr[st], X01 = xr[st + 1], X10 = xr[sb], X11 = xr[sb + 1];
                                const v4f G00 = *reinterpret_cast<const v4f*>(&L.G4s[st * 4]);
                                const v4f G01 = *reinterpret_cast<const v4f*>(&L.G4s[(st + 1) * 4]);   // (N, -, V, -)
                                const v2f G10 = *reinterpret_cast<const v2f*>(&L.G4s[sb * 4]);         // (N, H)
                                const float G11x = L.G4s[(sb + 1) * 4];
                                const float fe = 1.0f - fw[j], fs = 1.0f - fn[j];
                                // |sum_t w_t s'_t|^2, separable in the x weights (e, w) and the y weights (s, n)
                                const float ee = fe * fe, ww = fw[j] * fw[j], ew2 = 2.0f * (fe * fw[j]);
                                const float A = __builtin_fmaf(ee, G00.x, __builtin_fmaf(ww, G01.x, ew2 * G00.y));   // top row
                                const float Bq = __builtin_fmaf(ee, G10.x, __builtin_fmaf(ww, G11x, ew2 * G10.y));   // bottom row
                                const float Cq = __builtin_fmaf(ee, G00.z, __builtin_fmaf(ww, G01.z, (fe * fw[j]) * G00.w));   // cross rows
                                const float Q = __builtin_fmaf(fs * fs, A, __builtin_fmaf(fn[j] * fn[j], Bq, (2.0f * (fs * fn[j])) * Cq));
                                const float XW = __builtin_fmaf(fs * fe, X00, __builtin_fmaf(fs * fw[j], X01,
                                                 __builtin_fmaf(fn[j] * fe, X10, (fn[j] * fw[j]) * X11)));
                                q = __builtin_fmaf(-2.0f, XW, Q) + rr;
                                if (sl[j] & SL_BORDER) {
                                    // a cell on the image border: the taps outside read zero, not mu (header)
                                    const float wx = ((sl[j] & SL_XLO) ? 0.0f : fe) + ((sl[j] & SL_XHI) ? 0.0f : fw[j]);
                                    const float wy = ((sl[j] & SL_YLO) ? 0.0f : fs) + ((sl[j] & SL_YHI) ? 0.0f : fn[j]);
                                    const float om = 1.0f - wx * wy;
                                    const float MW = __builtin_fmaf(fs * fe, L.Ms[st], __builtin_fmaf(fs * fw[j], L.Ms[st + 1],
                                                     __builtin_fmaf(fn[j] * fe, L.Ms[sb], (fn[j] * fw[j]) * L.Ms[sb + 1])));
                                    q = __builtin_fmaf(om, __builtin_fmaf(om, M2, -2.0f * (MW - rho)), q);
                                }
                            }
                            // q / sigma through the divide chain of geometry.hpp (bit-identical to the IEEE divide for finite operands)
                            cost[4 * h + j] = cost[4 * h + j] + (fabsf(q) < 1.0e30f ? div_core(q, sg.x, sg.y) : q * sg.y);
                        }
                    }
                }
            }

            CSTAMP(9)   // combine
            if (failmask != 0) {
                // Passes whose geometry does not fit the row tables (more than MAXB blocks or rows: long epipolar segments): their
                // planes directly, in the reference's form on the centred features -- per plane the four taps of every channel
                // group by 16-byte loads from the packed source, the pixel's centred reference features and the means from LDS.
                // Taps outside the image: weight 0 on a clamped address (s' absent), and the (1 - Win) mu term of the header.
                // About twice the work of a pass that fits.
                const v4f c0 = *reinterpret_cast<const v4f*>(&L.cst[0]), c1 = *reinterpret_cast<const v4f*>(&L.cst[4]);
                const float M2 = L.mu[72];
#pragma unroll 1
                for (int vh = 0; vh < V * NH; ++vh) {
                    if (!(failmask >> vh & 1u)) continue;
                    if (tid == 0) ++n_direct;
                    const int v = vh / NH, h = vh - v * NH;
                    ViewXform xf;
#pragma unroll
                    for (int i = 0; i < 9; ++i) xf.kr[i] = L.xf[v * 12 + i];
#pragma unroll
                    for (int i = 0; i < 3; ++i) xf.kt[i] = L.xf[v * 12 + 9 + i];
                    xf.separate = ca.a.blas_mode;
                    float t2a, t2b, t2c;
                    ray_term2(xf, ray[0], ray[1], ray[2], t2a, t2b, t2c);
                    const float* srcf = reinterpret_cast<const float*>(ca.packed + ((size_t)b * V + v) * (NPL + 2) * HW);
#pragma unroll 1
                    for (int j = 0; j < 4; ++j) {
                        const int k = 64 * h + 4 * tq + j;
                        float ix, iy, fwj, fnj;
                        plane_sample_pos_fast(xf, t2a, t2b, t2c, L.dcl[k], c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, ix, iy);
                        int cellj = cell_of(ix, iy, W, H, fwj, fnj);
                        if (k >= D || !xlive) cellj = NO_CELL;
                        float q = (__builtin_fmaf(2.0f, rho, rr) + M2) + (fwj + fnj) * 0.0f;   // (no tap inside the image)
                        if (cellj != NO_CELL) {
                            const int cxx = cell_x(cellj), cyy = cell_y(cellj);
                            const bool x0in = cxx >= 0, x1in = cxx + 1 < W, y0in = cyy >= 0, y1in = cyy + 1 < H;
                            const float fe = 1.0f - fwj, fs = 1.0f - fnj;
                            const float wnw = x0in && y0in ? fs * fe : 0.0f, wne = x1in && y0in ? fs * fwj : 0.0f;
                            const float wsw = x0in && y1in ? fnj * fe : 0.0f, wse = x1in && y1in ? fnj * fwj : 0.0f;
                            const float om = (x0in && x1in && y0in && y1in) ? 0.0f : 1.0f - (((wnw + wne) + wsw) + wse);
                            const int xa = max(cxx, 0), xb = min(cxx + 1, W - 1), ya = max(cyy, 0), yb = min(cyy + 1, H - 1);
                            const int i00 = (ya * W + xa) * 4, i01 = (ya * W + xb) * 4, i10 = (yb * W + xa) * 4, i11 = (yb * W + xb) * 4;
                            float part = 0.0f;
#pragma unroll 1
                            for (int g = 0; g < NPL; ++g) {
                                const float* sg4 = srcf + (size_t)g * HW * 4;
                                const v4f t00 = *reinterpret_cast<const v4f*>(sg4 + i00), t01 = *reinterpret_cast<const v4f*>(sg4 + i01);
                                const v4f t10 = *reinterpret_cast<const v4f*>(sg4 + i10), t11 = *reinterpret_cast<const v4f*>(sg4 + i11);
                                const v4f u4 = *reinterpret_cast<const v4f*>(&L.mu[min(4 * g, 68)]);
                                v4f r4;   // channels 4 g .. 4 g + 3 of pixel n in the B-operand image (chunk g >> 2, lane slice g & 3)
                                if (g < 4 * NCH) r4 = *reinterpret_cast<const v4f*>(&L.Rs[(((g >> 2) * 4 + (g & 3)) * 16 + n) * 4]);
                                else {
                                    const float* rt = &L.Rs[Lds::RS_TAIL + ((g - 4 * NCH) * 4) * 16 + n];
                                    r4 = v4f{rt[0], rt[16], rt[32], rt[48]};
                                }
#pragma unroll
                                for (int i = 0; i < 4; ++i) {
                                    float val = t00[i] * wnw;
                                    val = __builtin_fmaf(t01[i], wne, val);
                                    val = __builtin_fmaf(t10[i], wsw, val);
                                    val = __builtin_fmaf(t11[i], wse, val);
                                    const float diff = (val - r4[i]) - om * u4[i];
                                    part = __builtin_fmaf(diff, diff, part);
                                }
                            }
                            q = part + (fwj + fnj) * 0.0f;
                        }
                        const float cj = fabsf(q) < 1.0e30f ? div_core(q, c1.z, c1.w) : q * c1.w;
#pragma unroll
                        for (int jj = 0; jj < NC; ++jj) cost[jj] = cost[jj] + (jj == 4 * h + j ? cj : 0.0f);
                    }
                }
                LDS_BARRIER();   // (every wave is done with the block's reference features: the next block may overwrite them)
            }
            // ---- epilogue: cost store, log-softmax over D, expectation ----------------------------------------------------
            // (buffer stores: one 32-bit lane offset, the plane as the scalar offset: plane 64 h + 16 wave + 4 kq + j)
            const int ovoff = xlive ? (4 * kq * HW + p) * 4 : OOB;
            float* const cost_out = ca.a.cost_out;
            float* const logp_out = ca.a.logp_out;
            float* const depth_out = ca.a.depth_out;
            if (cost_out) {
                const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc((void*)(cost_out + (size_t)b * D * HW), 0, D * HW * 4, 0x00020000);
#pragma unroll
                for (int j = 0; j < NC; ++j)   // (planes beyond D lie beyond the descriptor: dropped)
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, cost[j]), rc, ovoff, (64 * (j >> 2) + 16 * wave + (j & 3)) * HW * 4, CORR_STORE_AUX);
            }
            if (logp_out || depth_out) {
                // per wave: max, sum exp, sum d exp over its planes of pixel n; merged over the four waves through LDS
                float mx = -INFINITY;
#pragma unroll
                for (int j = 0; j < NC; ++j)
                    if (64 * (j >> 2) + 4 * tq + (j & 3) < D) mx = fmaxf(mx, cost[j]);
                mx = fmaxf(mx, __shfl_xor(mx, 16));
                mx = fmaxf(mx, __shfl_xor(mx, 32));
                float ssum = 0.0f, esum = 0.0f;
#pragma unroll
                for (int j = 0; j < NC; ++j) {
                    const int k = 64 * (j >> 2) + 4 * tq + (j & 3);
                    const float ek = k < D ? exp_nonpos(cost[j] - mx) : 0.0f;
                    ssum = ssum + ek;
                    esum = __builtin_fmaf(L.dcl[k], ek, esum);
                }
                ssum = ssum + __shfl_xor(ssum, 16); ssum = ssum + __shfl_xor(ssum, 32);
                esum = esum + __shfl_xor(esum, 16); esum = esum + __shfl_xor(esum, 32);
                if (kq == 0) *reinterpret_cast<v4f*>(&L.red[(wave * 16 + n) * 4]) = v4f{mx, ssum, esum, 0.0f};
                CSTAMP(10)   // epilogue: stores, partial softmax
                LDS_BARRIER();
                float M = -INFINITY;
#pragma unroll
                for (int w = 0; w < 4; ++w) M = fmaxf(M, L.red[(w * 16 + n) * 4]);
                float S = 0.0f, E = 0.0f;
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    const v4f part = *reinterpret_cast<const v4f*>(&L.red[(w * 16 + n) * 4]);
                    // (a wave whose planes all lie beyond D: max -inf, sums 0)
                    const float sc = part.x == -INFINITY ? 0.0f : exp_nonpos(part.x - M);
                    S = __builtin_fmaf(part.y, sc, S);
                    E = __builtin_fmaf(part.z, sc, E);
                }
                const float ls = logf(S);
                if (logp_out) {
                    const __amdgpu_buffer_rsrc_t rl = __builtin_amdgcn_make_buffer_rsrc((void*)(logp_out + (size_t)b * D * HW), 0, D * HW * 4, 0x00020000);
#pragma unroll
                    for (int j = 0; j < NC; ++j)
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, (cost[j] - M) - ls), rl, ovoff,
                                                              (64 * (j >> 2) + 16 * wave + (j & 3)) * HW * 4, CORR_STORE_AUX);
                }
                if (depth_out && xlive && tq == 0) depth_out[(size_t)b * HW + p] = E / S;
            }
            CSTAMP(11)   // barrier + merge + stores
        }   // pixel blocks of the item
    }   // items
#ifdef CORR_STAMPS
    if (lane == 0)
        for (int i = 0; i < 12; ++i) atomicAdd(reinterpret_cast<unsigned long long*>(KARG(int*, queue) + 16) + i, stamp_acc[i]);
#endif

    // the last workgroup to leave zeroes the queue counters: the next call on this workspace needs no clearing launch
    if (tid == 0) {
        int* queue = KARG(int*, queue);
        if (n_direct) atomicAdd(&queue[CORR_DIRECT_SLOT], n_direct);
        const int done = atomicAdd(&queue[CORR_DONE_SLOT], 1);
        if (done == (int)gridDim.x - 1) {
            const int nd = atomicAdd(&queue[CORR_DIRECT_SLOT], 0);
            for (int q = 0; q < 8; ++q) queue[q] = 0;
            queue[CORR_DONE_SLOT] = 0;
            queue[CORR_DIRECT_SLOT] = 0;
            if (npack > 0) {
                int* pc = KARG(int*, pack_ctr);
                for (int i = 0; i < 2 * KARG(int, a.B); ++i) pc[i] = 0;
            }
            queue[CORR_DIRECT_LAST_SLOT] = nd;   // diagnostics: pixel blocks of this call evaluated directly
        }
    }
}

template <int NPL, int NH>
hipError_t launch_inst(const SweepArgs& a, const float4* packed, const float* mu_tab, int* queue, int tiles_x, int tiles, int npack, int* pack_ctr,
                       hipStream_t stream) {
    auto kern = sweep_corr_kernel<NPL, NH>;
    // persistent grid: as many workgroups as the chip holds at once (registers and LDS decide: asked once per
    // instantiation and device), a multiple of 8; fewer when there is less work
    static int per_cu[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (per_cu[dev] == 0) {
        int nbk = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nbk, kern, 256, 0) != hipSuccess || nbk <= 0) nbk = 2;
        per_cu[dev] = nbk > 5 ? 5 : nbk;
    }
    long long nblk = ((long long)sweep_device_cus() * per_cu[dev] + 7) & ~7ll;
    CorrArgs ca;
    ca.a = a; ca.packed = packed; ca.mu_tab = mu_tab; ca.queue = queue; ca.tiles_x = tiles_x; ca.ntile = tiles;
    ca.npack = npack; ca.pack_ctr = pack_ctr;
    // small problems: one pixel block per item, so that every CU gets work
    ca.spi = (long long)tiles * a.B < CORR_SPI1_BELOW * nblk ? 1 : 4;
    const long long need = 8ll * ((tiles + 7) / 8) * (4 / ca.spi) * a.B;   // a workgroup per item of the largest XCD band, times 8
    if (need <= CORR_ONE_EACH_X * nblk) nblk = need;
    hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(256), 0, stream, ca);
    return hipGetLastError();
}

template <int NPL>
hipError_t launch_by_npl(int npl, const SweepArgs& a, const float4* packed, const float* mu_tab, int* queue, int tiles_x, int tiles,
                         int npack, int* pack_ctr, hipStream_t stream) {
    if (npl == NPL)
        return a.D <= 64 ? launch_inst<NPL, 1>(a, packed, mu_tab, queue, tiles_x, tiles, npack, pack_ctr, stream)
                         : launch_inst<NPL, 2>(a, packed, mu_tab, queue, tiles_x, tiles, npack, pack_ctr, stream);
    if constexpr (NPL > 1) return launch_by_npl<NPL - 1>(npl, a, packed, mu_tab, queue, tiles_x, tiles, npack, pack_ctr, stream);
    return hipErrorInvalidValue;
}

}  // namespace

// shapes the kernel is built for: L2, D <= 128, C <= 72, at most 8 source views (one instantiation per ceil(C/4): the reference features of a
// pixel and the texel features of a block live in registers)
constexpr int CORR_MAX_NPL = 18;
bool sweep_corr_supports(const SweepArgs& a) {
    const int npl = (a.C + 3) / 4;
    const long long hw = (long long)a.H * a.W;
    return a.metric == 0 && a.D <= 128 && npl <= CORR_MAX_NPL && a.V <= CORR_MAXV && a.W <= 32760 && a.H <= 32760 &&
           hw * a.D * 4 < (1ll << 31) && hw * a.C * 4 < (1ll << 31) && hw * (npl + 2) * 16 < (1ll << 31) && hw * 12 < (1ll << 31);
}

// Launches the pre-pass and the sweep kernel.  NCHW entry: channel means and pack kernel (launch_pack_c4), then the sweep
// kernel -- or, on request, the channel means and a sweep kernel that packs the centred source itself, batch item b + 1
// under the sweep of item b.  Packed entry: the sweep kernel alone.
hipError_t launch_sweep_corr(const SweepArgs& a, void* workspace, hipStream_t stream, bool packed_ready) {
    const int tiles_x = (a.W + 15) / 16, tiles_y = (a.H + 3) / 4, tiles = tiles_x * tiles_y;
    float4* packed = reinterpret_cast<float4*>(static_cast<char*>(workspace) + sweep_ws_flag_bytes(a.B, a.H, a.W));
    int* queue = reinterpret_cast<int*>(static_cast<char*>(workspace) + sweep_ws_flag_only_bytes(a.B, a.H, a.W));
    const float* mu_tab = reinterpret_cast<const float*>(static_cast<char*>(workspace) + sweep_ws_stats_offset(a.B, a.V, a.C, a.H, a.W));
    int npack = 0;
    int* pack_ctr = nullptr;
    if (!packed_ready) {
        const long long items = (long long)a.V * ((a.H * a.W + 255) / 256);
        // (measured slower than the pack kernel in front -- profiles/r04_ab/fuse_pack_*.txt, DESIGN.md section 3 -- and so
        //  only on request: PDEPTH_CORR_FUSE_PACK=1)
#ifdef PDEPTH_LAB
        const char* fuse = getenv("PDEPTH_CORR_FUSE_PACK");
#else
        const char* fuse = nullptr;   // (the product library has no environment switches)
#endif
        if (fuse && fuse[0] == '1' && sweep_ws_holds_pack_counters(a.B, a.H, a.W) && items < (1ll << 30)) {
            hipError_t e = launch_stats_only(a, workspace, stream);
            if (e != hipSuccess) return e;
            npack = (int)items;
            pack_ctr = sweep_ws_pack_counters(a, workspace);
        } else {
            hipError_t e = launch_pack_c4(a, workspace, stream, /*centre=*/true);
            if (e != hipSuccess) return e;
        }
    }
    return launch_by_npl<CORR_MAX_NPL>((a.C + 3) / 4, a, packed, mu_tab, queue, tiles_x, tiles, npack, pack_ctr, stream);
}

}  // namespace pdepth
