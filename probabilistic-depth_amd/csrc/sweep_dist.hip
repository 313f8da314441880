// Plane sweep in the distance form of the L2 cost, one workgroup per block of 16 pixels: the default path of
// pdepth_sweep_{cost,dpv}_f32 for the L2 metric (C <= 72, D <= 128, at most 8 source views).
//
//   what is computed.  est_swp_volume_v4 (warping/homography.py:98-135) evaluates, per pixel p and plane k,
//           cost = | sum_t w_t s_t - r |^2 / sigma     (img_dis_L2_pard :80-82; t = the four bilinear taps :197, zero outside)
//       and with sum_t w_t = 1 (dist_layout.hpp)
//           | sum_t w_t s_t - r |^2 = sum_t w_t Y_t - Q,    Y_t = |s_t - r|^2,    Q = sum_{t<t'} w_t w_t' |s_t - s_t'|^2.
//       Y for 16 neighbouring pixels x 16 consecutive texels of a source row is one 16 x 16 x K matrix product on fp16
//       high / low parts of the centred, power-of-two scaled features (v_mfma_f32_16x16x32_f16, fp32 accumulation: 7
//       instructions per block at C = 67, products exact, the sum as accurate as the fp32 matrix instruction's --
//       tools/mb_split16.hip); Q = five squared neighbour differences per source cell from the pre-pass (pack_dist.hip).
//       The source image carries a ring of zero-feature texels: no border cases.
//
//   how.  The 64 (128) planes of the 16 pixels are spread over the 256 threads of a workgroup -- thread (pixel n, tq) owns
//       planes 4 tq .. 4 tq + 3 of each group of 64 (a PASS = one source view x one group of 64 planes):
//         positions   bit-faithful sample positions (geometry.hpp; scalar fp32 instructions: wave_util.hpp says why not packed);
//         row table   per source row the run of texels any sample touches: LDS min / max, rows indexed modulo 64; barrier;
//                     EVERY wave cuts the runs into blocks of 16 texels for itself (a prefix sum over the rows in registers:
//                     no serial phase, no block list in LDS; a sample finds its slot with two ds_bpermute);
//         Y           wave w multiplies blocks w, w + 4, ...: a block's texel operands are 2 NCHK + 1 16-byte loads per lane
//                     whose only per-block address part is a scalar offset; one register set, every chunk refilled with the
//                     next block's right behind its last multiplication; Y[pixel][slot] to LDS; the
//                     Q records of the block's 16 cells straight from memory to LDS (buffer_load ... lds); barrier;
//         combine     per (pixel, plane): 4 Y values, 5 Q values, the bilinear weights;
//         epilogue    cost store; log-softmax over D and E[d]: per wave partial (max, sum, sum d) of each pixel, merged
//                     across the four waves through LDS (one barrier per pixel block).
//       A pass whose planes need more than MAXB blocks or 64 rows (extreme poses), and every pass of a batch item whose
//       statistics put it outside the domain of the distance form (guard below), is evaluated directly by the same
//       workgroup behind the view loop, in the reference's own form on the packed features.
//   scheduling.  Persistent workgroups pull 16x4 tiles from per-XCD queues (balanced half-bands, stealing); the last
//       workgroup to leave zeroes the queue counters: a call on an already packed source is this one launch.
#include <hip/hip_runtime.h>

#include <atomic>
#include <climits>
#include <cstdlib>

#include "dist_layout.hpp"
#include "geometry.hpp"
#include "kernels.hpp"
#include "pick.hpp"
#include "wave_util.hpp"

#include "sweep_dist_knobs.hpp"

namespace pdepth {

namespace {

using namespace wv;
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef int v2i __attribute__((ext_vector_type(2)));

constexpr int DIST_MAXV = 8;   // source views whose homography terms a workgroup keeps in LDS

struct DistArgs {
    SweepArgs a;
    const char* packed;
    const float* stats;
    int* queue;
    int* qcnt;      // the eight per-XCD queue counters, qstride ints apart (in the workspace's tile-flag ints, which this kernel family does not use otherwise)
    int qstride;
    int tiles_x, ntile;
    int one_each;   // no queue: workgroup i runs item i (launch_inst says when)
    int nblk;       // workgroups of the launch
    int nonce;  // 1 .. 2047, another one per launch: tags the diagnostics count of this call (kernels.hpp: DIST_NONCE_SLOT)
};
#define KARG(type, field) kernarg_at<type>(offsetof(DistArgs, field))


// exp(x) for x <= 0 (softmax terms): the hardware 2^t on the rounded product t = x log2(e).  The product's rounding error
// (2^-24 |t|) is a relative error |t| ln2 2^-24 of the result: 4e-7 for the terms within a factor 1000 of the largest one,
// which are the ones that carry a depth map or a normaliser; geometry.hpp's exp_nonpos (12 instructions) keeps 1.5 ulp
// for every term (DIST_EXACT_EXP restores it).
__device__ __forceinline__ float dist_exp(float x) {
    if (DIST_EXACT_EXP) return exp_nonpos(x);
    return __builtin_amdgcn_exp2f((x < -1000.0f ? -1000.0f : x) * 1.44269502162933349609375f);   // (a NaN passes through, as in exp_nonpos)
}

template <int MAXB, int NAC>
struct __attribute__((aligned(16))) DistLds {
    static constexpr int XSTRIDE = MAXB * 16 + 4;   // floats per pixel of the Y buffer (stride / 4 odd: conflict-free b128 stores)
    float Ys[16 * XSTRIDE + DIST_YSKEW];   // Y[pixel][slot]; the rows of pixels 8 .. 15 begin DIST_YSKEW floats later (their slots differ from
                                           // those of pixels 0 .. 7 by a source row = a multiple of 16: the same LDS banks without the skew)
    float Qs[(MAXB + 3) / 4 * 256 + 8];   // Q record (Dx0, Dy0, Dd, Dx1) per slot (moved in groups of four blocks: whole groups)
    _Float16 Bs[NAC * 4 * 16 * 8];   // pixel-side operands of the block's 16 pixels: [chunk][kq][pixel][8], -2 x (high | low) parts
    float rp[4 * 16 * 2];        // per wave and pixel: partial |r'|^2, |r|^2 (scaled)
    float mus[dist::MAX_C + 8];  // channel means x scale of the batch item in work
    float xf[DIST_MAXV * 12];    // per view: K@R (9), K@t (3)
    float cst[8];                // cx, cy, 1/cx, 1/cy, W/2, H/2, scale, 2^(-2e)/sigma
    float red[4 * 16 * 4];       // epilogue exchange: (max, sum, sum d) per wave and pixel
    float dcl[128];              // depth candidates
    int ctab[2][64 * 2];         // per cell row (modulo 64): (min, max) x0; two sets, alternating by pass
    int ired[2][2];              // min / max cell row of the pass; two sets
    int brow[MAXB + 2];          // per block of the pass: its row (written alike by every wave, read back by the same wave)
    int item[2];                 // work item: current / next
    int iflag;                   // the batch item in work: 1 = fp16 overflow in the pack, 2 = outside the domain (guard)
    unsigned char wide[64];      // per batch item: pixel blocks are 16x1 (else 8x2)
};

// NCHK = chunks of 32 channels (dist_layout.hpp); NH = groups of 64 planes (ceil(D / 64): 1 or 2), each a pass of its own per
// view: thread (n, tq) owns planes 64 h + 4 tq .. + 3 of pass h.
//
// Everything a pass needs beyond its own registers lives in LDS or is re-read from the kernel-argument segment where it is
// used (KARG): the kernel keeps no spilled scalar registers (tests/test_isa_guard.py).  Round 5's build of this kernel carried
// 120 of them -- 14 % of its vector instructions were v_readlane / v_writelane traffic.
template <int NCHK, int NH>
__global__ __launch_bounds__(256, NH == 1 ? DIST_OCC1 : DIST_OCC2) void sweep_dist_kernel(DistArgs da) {
    constexpr int NAC = 2 * NCHK + 1;              // operand chunks of a texel / pixel: high[NCHK], low[NCHK], tail
    constexpr int MP = NCHK + 1;                   // rounds of the cooperative reference load: 16 channel PAIRS per round (the last: the tail's 4)
    constexpr int MAXB = NH == 1 ? DIST_MAXB1 : DIST_MAXB2;
    constexpr int NC = 4 * NH;                     // planes (costs) per thread
    constexpr int QPL = 8 * NCHK + 3;              // the Q plane
    constexpr int NS = NH == 1 ? DIST_SETS1 : DIST_SETS2;   // texel operand register sets = blocks of a wave in flight
    typedef DistLds<MAXB, NAC> Lds;
    constexpr int XSTRIDE = Lds::XSTRIDE;
    __shared__ Lds L;
#ifdef DIST_STAMPS   // (from the workgroup's first instruction: its start-up counts as "queue")
    unsigned long long stamp_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long stamp_t = __builtin_amdgcn_s_memtime();
#endif
    if (poison_on_foreign_layout(da.a, da.queue, LAYOUT_DIST16)) return;
    const int tid = threadIdx.x;

    {
        const float* dc = KARG(const float*, a.d_candi);
        const int D = KARG(int, a.D);
        for (int k = tid; k < 64 * NH; k += 256) L.dcl[k] = dc[min(k, D - 1)];
    }
    if (tid < 128) { L.ctab[0][tid] = (tid & 1) ? INT_MIN : INT_MAX; L.ctab[1][tid] = (tid & 1) ? INT_MIN : INT_MAX; }
    if (tid < 64) L.wide[tid] = 1;
    if (tid < 2) { L.ired[tid][0] = INT_MAX; L.ired[tid][1] = INT_MIN; }
    __syncthreads();
    // Shape of the pixel blocks of a batch item: 16x1 where the epipolar lines of view 0 run along the source rows (a
    // rectified pair: the 16 pixels of a row share two source rows), else 8x2 (pick.hpp; any choice is correct).
    if ((int)(tid >> 2) < min(KARG(int, a.B), 64) && epipolar_probe_is_steep(da.a, tid >> 2, tid & 3)) L.wide[tid >> 2] = 0;

    // ---- work queue (per XCD) -------------------------------------------------------------------------------------------
    // Workgroups are dealt round-robin over the 8 XCDs; XCD q owns the tiles of band q (its own queue counter), so that
    // neighbouring tiles, whose source texels overlap, meet in that XCD's L2.  A workgroup whose queue is exhausted takes
    // items of the others.
    const int xcd = blockIdx.x & 7;
    if (blockIdx.x == 0 && tid == 0) KARG(int*, queue)[DIST_NONCE_SLOT] = KARG(int, nonce);   // (diagnostics: whose count DIST_DIRECT_LAST_SLOT holds)
    // the tags of the fused statistics (pack_dist.hip: fused_stats) are cleared for the next call on this workspace: a captured graph
    // replays the pack kernel with the same tag (workgroup 1; nobody reads the tags during a sweep)
    if (blockIdx.x == 1 % gridDim.x) {
        float* st = const_cast<float*>(KARG(const float*, stats));
        for (int i = tid; i < KARG(int, a.B) * STATS_VAR; i += 256) reinterpret_cast<int*>(st + (size_t)(i / STATS_VAR) * STATS_STRIDE + STATS_READY)[i % STATS_VAR] = 0;
    }
    auto band_tiles_of = [&](int q) { const int nt = KARG(int, ntile); return (nt >> 3) + (q < (nt & 7) ? 1 : 0); };
    auto band_first_of = [&](int q) {
        const int nt = KARG(int, ntile), qq = nt >> 3, rr8 = nt & 7;
        return q < rr8 ? q * (qq + 1) : rr8 * (qq + 1) + (q - rr8) * qq;
    };
    // items of queue q: the pixel blocks of its tiles, four per tile, in the order of the tiles.  The ~96 workgroups of an XCD
    // pop consecutive items within a microsecond of one another: at any time they cover a compact patch of ~24 neighbouring
    // tiles, and the four blocks of a tile -- which share nearly all their source texels -- are in work TOGETHER, sharing
    // them through the XCD's L2 at once.  (Round 5 handed out whole tiles, a workgroup running a tile's four blocks one
    // after the other, 9 us apart: L2 hit rate 67 %, with single blocks 84 % and half the misses -- profiles/r06_ab/.)
    auto items_of = [&](int q) { return 4 * band_tiles_of(q) * KARG(int, a.B); };
    int own_done = 0;   // (ints, not bools: a uniform bool that lives across the kernel is kept as a 64-bit lane mask)
    auto steal = [&]() -> int {   // (queue << 28) | index in the queue, or -1: every queue is exhausted
        int* qcnt = KARG(int*, qcnt);
        const int qs = KARG(int, qstride);
#pragma unroll 1   // (unrolled, the seven queue numbers and their shifted forms were fourteen spilled scalars of the whole kernel)
        for (int j = 1; j < 8; ++j) {
            const int q = (opaque_s(xcd) + j) & 7, nq = items_of(q);
            if (__hip_atomic_load(&qcnt[q * qs], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= nq) continue;
            const int got = atomicAdd(&qcnt[q * qs], 1);
            if (got < nq) return (q << 28) | got;
        }
        return -1;
    };
    auto resolve = [&](int got) -> int {
        const int n_own = items_of(xcd);
        if (!own_done && got < n_own) return (opaque_s(xcd) << 28) | got;
        own_done = 1;
        return steal();
    };
    // floor(nn / dd) for 0 <= nn < 2^22, 0 < dd (an integer divide costs ~40 dependent instructions)
    auto fdiv = [](int nn, int dd) { return (int)(((float)nn + 0.5f) * __builtin_amdgcn_rcpf((float)dd)); };
    auto decode = [&](int item, int& b_, int& tx_, int& ty_, int& sub_) {
        const int ntile = KARG(int, ntile), tiles_x = KARG(int, tiles_x), H = KARG(int, a.H);
        const int qq = ntile >> 3, rr8 = ntile & 7;
        const bool small_idx = (long long)ntile * 4 * KARG(int, a.B) < (1ll << 22);
        const int tiles_y_ = (H + 3) / 4;
        const int q_ = item >> 28, iq = item & 0x0fffffff, band_tiles = band_tiles_of(q_);
        // index of the tile in the queue's order (batch item by batch item), and the pixel block of the tile
        const int tidx = iq >> 2;
        sub_ = iq & 3;
        b_ = small_idx ? fdiv(tidx, band_tiles) : tidx / band_tiles;
        const int ti = tidx - b_ * band_tiles;
        int tile = band_first_of(q_) + ti;
        if (DIST_BANDS == 8 && rr8 == 0 && tiles_y_ % 8 == 0 && tiles_y_ * tiles_x == ntile) {
            // XCD q owns band q of the image's 8 (twice the rows of the half-bands below: the source rows a band reaches
            // beyond itself are fetched by 8 L2s per item, not 16)
            const int hb_rows = tiles_y_ / 8;
            const int cc = small_idx ? fdiv(ti, hb_rows) : ti / hb_rows, r_ = ti - cc * hb_rows;
            const int col = DIST_COL_ALT ? ((cc & 1) ? tiles_x - 1 - (cc >> 1) : (cc >> 1)) : cc;
            tile = (q_ * hb_rows + r_) * tiles_x + col;
        } else if (rr8 == 0 && tiles_y_ % 16 == 0 && tiles_y_ * tiles_x == ntile) {
            // XCD q owns half-bands q and 8 + q of the image's 16: on a forward motion the cost of a tile grows with its
            // distance from the image centre, and this way every XCD gets the same mix; the heavier half first and, inside
            // a half, columns left to right.  (Any static partition is valid: dry queues steal.)
            const int hb_rows = tiles_y_ / 16, half_tiles = hb_rows * tiles_x;
            const int second = ti >= half_tiles ? 1 : 0, tih = ti - second * half_tiles;
            const int hbi = (q_ < 4) == (second == 0) ? q_ : 8 + q_;
            const int cc = small_idx ? fdiv(tih, hb_rows) : tih / hb_rows, r_ = tih - cc * hb_rows;
            const int col = DIST_COL_ALT ? ((cc & 1) ? tiles_x - 1 - (cc >> 1) : (cc >> 1)) : cc;
            tile = (hbi * hb_rows + r_) * tiles_x + col;
        } else if (rr8 == 0 && qq % tiles_x == 0) {   // the band is a whole number of tile rows: column by column
            const int band_rows = qq / tiles_x, tc = small_idx ? fdiv(ti, band_rows) : ti / band_rows;
            tile = (q_ * band_rows + (ti - tc * band_rows)) * tiles_x + tc;
        }
        ty_ = small_idx ? fdiv(tile, tiles_x) : tile / tiles_x;
        tx_ = tile - ty_ * tiles_x;
    };
    // (when the grid covers every item -- small problems -- workgroup i takes item i of its XCD's band: no atomics.  Re-read from
    //  the kernel arguments where it is used: a flag that lives across the kernel is a scalar register pair spilled)
#define ONE_EACH (KARG(int, one_each) != 0)
    __syncthreads();

    // Persistent grid: the workgroups a CU starts with would run their first passes in lockstep -- every phase of all three at
    // once on the same pipes; the second and third begin 5 and 11 us later (they drift apart by themselves within a few
    // passes; -0.9 % / -1.8 % of the headline launch, profiles/r05_ab/staggered_start.txt).  Not where a workgroup runs one item.
    if (DIST_STAGGER && !ONE_EACH) {
        const int slot = ((int)blockIdx.x >> 3) / 32 % 3;
        for (int i = 0; i < slot * DIST_STAGGER; ++i) __builtin_amdgcn_s_sleep(100);
    }
    // loop-carried scalars of the kernel in ONE register: bit 0 = the slot of L.item in use, bit 1 = the set of row-table arrays of
    // the pass (alternating), bits 2.. = 1 + the batch item whose tables (means, homography terms, camera constants) are in LDS
    int state = 0;
    int n_direct = 0;     // (thread 0) pixel blocks evaluated directly
    // The queue runs one item ahead: (thread 0) the atomic that pops item i + 1 is issued when item i starts, its result is
    // resolved and published in LDS in front of the LAST barrier of item i -- the workgroup goes from one pixel block to
    // the next without a barrier of the queue's own.
    int got_own = (tid == 0 && !ONE_EACH) ? atomicAdd(&KARG(int*, qcnt)[xcd * KARG(int, qstride)], 1) : (int)(blockIdx.x >> 3);
    // resolve_next(): BEFORE the block's stores are issued -- the result of the atomic is waited for with a vmcnt, and memory
    // operations complete in issue order: behind the stores that wait is the stores' whole latency, for thread 0's wave and, at
    // the next barrier, for the workgroup (2 us per pixel block when it was there).
    int next_item = -1;
    // ("thread 0" asked of a value the optimiser cannot see through: the lane mask of `tid == 0` is not kept across the kernel)
    auto resolve_next = [&]() { if (opaque_v((int)threadIdx.x) == 0) next_item = ONE_EACH ? -1 : resolve(got_own); };
    auto publish_next = [&]() {   // (thread 0, in front of a workgroup barrier)
        if (opaque_v((int)threadIdx.x) == 0) {
            L.item[state & 1] = next_item;
            if (!ONE_EACH && !own_done) got_own = atomicAdd(&KARG(int*, qcnt)[xcd * KARG(int, qstride)], 1);
        }
    };
    if (tid == 0) {
        L.item[0] = ONE_EACH ? (got_own < items_of(xcd) ? (xcd << 28) | got_own : -1) : resolve(got_own);
        if (!ONE_EACH && !own_done) got_own = atomicAdd(&KARG(int*, qcnt)[xcd * KARG(int, qstride)], 1);
    }
    PDEPTH_LDS_BARRIER();   // the first item is published

    for (;;) {
        // (a plain LDS read: behind the barrier's memory clobber it cannot be hoisted.  Through a volatile generic pointer --
        //  round 5 -- it was a FLAT load, and a flat load is waited for with vmcnt(0): every item began by waiting out the
        //  previous pixel block's stores, 2 us per item)
        const int item = __builtin_amdgcn_readfirstlane(L.item[state & 1]);
        state ^= 1;
        if (item < 0) break;
        DSTAMP(0)   // queue
        const int wave = opaque_s(__builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6));   // (per item: its multiples are not kept -- spilled -- across the kernel)
        int b, tx, ty, sub;
        decode(item, b, tx, ty, sub);
        const int H = KARG(int, a.H), W = KARG(int, a.W), V = KARG(int, a.V), C = KARG(int, a.C), D = KARG(int, a.D);
        const bool wide = b < 64 ? L.wide[b] != 0 : false;

        {
            if ((wide ? ty * 4 + sub : ty * 4 + 2 * (sub >> 1)) >= H) {   // the block lies below the image (uniform)
                resolve_next();
                publish_next();
                PDEPTH_LDS_BARRIER();
                continue;
            }
            // lane roles: in the vector phases thread (n, tq) owns pixel n of the block and planes 64 h + 4 tq .. + 3; in the
            // matrix phase lane (n, kq) of a wave feeds texel / pixel n and K slice kq.  (opaque: the optimiser otherwise
            // hoists every lane-derived invariant of the phases to the top of the kernel)
            const int tid = opaque_v((int)threadIdx.x), lane = tid & 63;   // (shadow the kernel's: re-derived per trip)
            const int n = lane & 15, kq = lane >> 4, tq = wave * 4 + kq;
            const int a32 = (lane ^ 32) << 2;   // (bperm_f: the lane of pixel n two K slices on)
            const int HW = H * W;
            int p;
            bool xlive;
            float ray[3], rv[MP][2];
            // the pixel's ray, and this thread's share of the block's reference features: channel pairs tq, tq + 16, ... of
            // pixel n (buffer loads: 32-bit offsets, the channel as the scalar offset; channels beyond C = 0); the tail's four
            // pairs are wave 0's
            {
                const __amdgpu_buffer_rsrc_t rray =
                    __builtin_amdgcn_make_buffer_rsrc((void*)(KARG(const float*, a.rays) + (size_t)b * 3 * HW), 0, 3 * HW * 4, 0x00020000);
                const __amdgpu_buffer_rsrc_t rref = __builtin_amdgcn_make_buffer_rsrc(
                    (void*)(KARG(const float*, a.ref) + (size_t)b * KARG(long long, a.ref_bstride)), 0, C * HW * 4, 0x00020000);
                const int x = wide ? tx * 16 + n : tx * 16 + 8 * (sub & 1) + (n & 7);
                const int y = wide ? ty * 4 + sub : ty * 4 + 2 * (sub >> 1) + (n >> 3);
                xlive = x < W && y < H;
                p = min(y, H - 1) * W + min(x, W - 1);
#pragma unroll
                for (int i = 0; i < 3; ++i) ray[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rray, p * 4, i * HW * 4, 0));
#pragma unroll
                for (int mm = 0; mm < MP; ++mm) {
                    // (the tail round's loads are issued by every wave -- beyond the descriptor in waves 1..3 -- : registers that
                    //  only wave 0 loads made the compiler wait, at the top of every trip, for the previous trip's stores)
                    const bool mine = mm < NCHK || wave == 0;
#pragma unroll
                    for (int i = 0; i < 2; ++i)
                        rv[mm][i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                            rref, mine && 2 * tq + i < C - 32 * mm ? ((2 * tq + i) * HW + p) * 4 : OOB, 32 * mm * HW * 4, 0));
                }
            }
            DSTAMP(1)   // item set-up, pixel loads issued
            // (the pixel loads above are in flight while the tables of a new batch item are built: where every workgroup runs ONE
            //  item -- the model-real shapes -- that is a memory round trip off a 13 us launch)
            // per BATCH item, for every wave: scaled channel means, the views' homography terms, the camera constants, the item's
            // flags (visible behind the barrier in front of the first block's centring).  Items come out of the queues batch item
            // by batch item: the tables are rebuilt a few times per launch, not once per tile.
            const bool new_b = b + 1 != state >> 2;
            if (new_b) {
                state = (state & 3) | ((b + 1) << 2);
                // (the lane's roles re-derived here: hoisted out of the item loop, the invariants of this rarely run block were
                //  spilled registers of the whole kernel)
                const int tid = opaque_v((int)threadIdx.x), lane = tid & 63;
                const float* st = KARG(const float*, stats) + (size_t)b * STATS_STRIDE;
                if (wave == 0) {
                    // the scale (dist_layout.hpp: the same function of the same numbers as in the pack kernel) and the guard
                    float am = st[STATS_AMAX + lane], sv = st[STATS_VAR + lane], sl_ = st[STATS_LAG + lane], m2 = st[lane] * st[lane];
                    if (lane + 64 < STATS_VAR) {
                        am = fmaxf(am, st[STATS_AMAX + lane + 64]); sv += st[STATS_VAR + lane + 64]; sl_ += st[STATS_LAG + lane + 64];
                        m2 += st[lane + 64] * st[lane + 64];
                    }
                    // (the largest and the smallest |depth candidate|: L.dcl repeats the last one beyond D)
                    float dhi = fabsf(L.dcl[lane]), dlo = -dhi;
                    if (NH == 2) { const float d2 = fabsf(L.dcl[lane + 64]); dhi = fmaxf(dhi, d2); dlo = fmaxf(dlo, -d2); }
                        // (xor shuffles whose lane arithmetic is part of the instruction -- ds_swizzle -- or hangs on the opaque lane
                    //  above: __shfl_xor's own, hoisted out of the item loop, was a spilled register of the kernel)
                    {
                        const int x32 = (lane ^ 32) << 2;
                        am = fmaxf(am, bperm_f(x32, am)); sv += bperm_f(x32, sv); sl_ += bperm_f(x32, sl_); m2 += bperm_f(x32, m2);
                        dhi = fmaxf(dhi, bperm_f(x32, dhi)); dlo = fmaxf(dlo, bperm_f(x32, dlo));
#define PDEPTH_XOR_STEP(M) { const int pat = ((M) << 10) | 0x1f; \
                             am = fmaxf(am, __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, am), pat))); \
                             sv += __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, sv), pat)); \
                             sl_ += __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, sl_), pat)); \
                             m2 += __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, m2), pat)); \
                             dhi = fmaxf(dhi, __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, dhi), pat))); \
                             dlo = fmaxf(dlo, __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, dlo), pat))); }
                        PDEPTH_XOR_STEP(16) PDEPTH_XOR_STEP(8) PDEPTH_XOR_STEP(4) PDEPTH_XOR_STEP(2) PDEPTH_XOR_STEP(1)
#undef PDEPTH_XOR_STEP
                    }
                    const int e = dist::scale_exponent(am);
                    const float sc = ldexpf(1.0f, e);
                    // Guard.  The rounding error of Y = N - 2 X + |r'|^2 is 2^-23 of the ENERGY of the centred features (sum_c var_c,
                    // trends across the image included: a constant per channel does not remove them), that of the reference's own
                    // form 2^-23 of the cost, i.e. of their spread at the distance of a sweep (sum_c lag_c).  Measured on features
                    // with trends (tools/dbg/dist_guard.py, depth against the CPU oracle): ratio 1.34 -> 3.6e-5 m, ratio 2.1 ..
                    // 2.4 -> 1.0e-4 .. 1.4e-4 m, where the reference's form keeps 3e-5.  An item is evaluated directly where the
                    // ratio exceeds DIST_GUARD_RATIO AND the energy is large enough against sigma for the difference to show
                    // (unit-variance features: 67; features whose energy is all spread gain nothing from the direct form).
                    const float sg = KARG(float, a.sigma);
                    const bool outside = sv > DIST_GUARD_RATIO * sl_ && sv * 10.0f > DIST_GUARD_ENERGY * fabsf(sg);
                    // Conditioning.  Whatever the form, an fp32 cost carries 2^-23 of its own size, and the expected depth moves by
                    // up to the candidates' range times that: where V (2 sum var + |mu|^2) / sigma -- the cost of a sample, inside
                    // the image and outside it -- times the range times 2^-23 exceeds PDEPTH_COND_LIMIT, two fp32 evaluations agree
                    // to 1e-4 m only if they round alike (the float32 reference itself is then up to 3e-4 m from the exact value:
                    // tests/test_soak_regressions.py).  Headline workload: 5.6e-5, config 5: 2.2e-4; the six soak cases: 5.8e-4
                    // .. 3.5e-3.  Such an item is left to the gather kernel, which rounds like the reference -- where the caller
                    // gave the NCHW source (below); a packed source has nothing to fall back to and takes the fast form.
                    const bool illcond = (float)KARG(int, a.V) * (2.0f * sv + m2) * (dhi + dlo) * 1.1920929e-7f > PDEPTH_COND_LIMIT * fabsf(sg);
                    if (lane == 0) {
                        L.cst[6] = sc; L.cst[7] = ldexpf(refined_rcp(sg), -2 * e);
                        const int pf = reinterpret_cast<const int*>(st + STATS_FLAGS)[0];
                        L.iflag = (pf != 0 ? 1 : 0) | (outside ? 2 : 0) | (illcond ? 4 : 0);
                    }
                    for (int c = lane; c < dist::MAX_C + 8; c += 64) L.mus[c] = st[c] * sc;
                }
                if (tid >= 128 && tid < 128 + V) {
                    const int v = tid - 128;
                    ViewXform xf;
                    make_view_xform(KARG(const float*, a.K) + b * 9, KARG(const float*, a.R) + ((size_t)b * V + v) * 9,
                                    KARG(const float*, a.t) + ((size_t)b * V + v) * 3, KARG(int, a.blas_mode), xf);
    #pragma unroll
                    for (int i = 0; i < 9; ++i) L.xf[v * 12 + i] = xf.kr[i];
    #pragma unroll
                    for (int i = 0; i < 3; ++i) L.xf[v * 12 + 9 + i] = xf.kt[i];
                }
                if (tid == 255) {
                    const float* const cxcy_ = KARG(const float*, a.cxcy);
                    const float cx = cxcy_[b * 2 + 0], cy = cxcy_[b * 2 + 1];
                    L.cst[0] = cx; L.cst[1] = cy; L.cst[2] = refined_rcp(cx); L.cst[3] = refined_rcp(cy);
                    L.cst[4] = (float)W / 2.0f; L.cst[5] = (float)H / 2.0f;
                }
            }
            bool item_ready = !new_b;
            // Routing (NCHW entry: the raw source is at hand): an item the guard or the conditioning flags is the gather kernel's,
            // launched behind this one for the items marked here (capi.hip); its pixel blocks are skipped.
            if (KARG(const float*, a.src) != nullptr) {
                if (!item_ready) { PDEPTH_LDS_BARRIER(); item_ready = true; }
                if ((__builtin_amdgcn_readfirstlane(L.iflag) & 7) != 0) {   // uniform (overflow of the fp16 range, guard, conditioning)
                    // (the item's first block marks it and counts ALL its blocks for the diagnostics: the count is merged into
                    //  the workspace by a compare-and-swap chain per workgroup at the end of the kernel -- one per pixel block was
                    //  0.9 ms of serialised atomics on a routed launch)
                    if (opaque_v((int)threadIdx.x) == 0 && tx == 0 && ty == 0 && sub == 0) {
                        reinterpret_cast<int*>(const_cast<float*>(KARG(const float*, stats)) + (size_t)b * STATS_STRIDE + STATS_FLAGS)[1] = 1;
                        n_direct += KARG(int, tiles_x) * (wide ? H : 2 * ((H + 1) >> 1));
                    }
                    resolve_next();
                    publish_next();
                    PDEPTH_LDS_BARRIER();
                    continue;
                }
            }
            float rr = 0.0f, zc = 0.0f;    // |r'|^2 and |r|^2 of the pixel, scaled (set with the first pass)
            bool centred = false;
            unsigned failmask = 0;         // uniform over the workgroup: half-passes (view v, plane group h, waves 0-1 | 2-3) left to the direct evaluation
            float cost[NC];
#pragma unroll
            for (int j = 0; j < NC; ++j) cost[j] = 0.0f;

            constexpr bool RETRY = NH == 2;   // (passes that do not fit are run again in halves: below)
            for (int v = 0; v < V; ++v) {
                // (NH = 1: the view's address formed once per view, as the straight-line pass always had it)
                const char* const srcv_v = RETRY ? nullptr : KARG(const char*, packed) + (size_t)((size_t)b * V + v) * dist::view_bytes(C, H, W);
                const int H_ = H, W_ = W, C_ = C, D_ = D;
#pragma unroll
                for (int h = 0; h < NH; ++h) {
                    if (!item_ready) { PDEPTH_LDS_BARRIER(); item_ready = true; }   // (the item's tables above are in LDS)
                    // A pass whose texel blocks do not fit the workgroup's LDS (nb > MAXB: the large windows of a forward motion at D >
                    // 64) is run again as two HALF-passes -- the planes of waves 0-1, then those of waves 2-3: half the planes
                    // touch about half the rows --, and only a half that still does not fit is left to the direct evaluation
                    // (config 5: 1 482 whole passes were, 0.1 of its 3.45 ms).  attempt 0: all planes; 1, 2: the halves.  (The
                    // positions are worked out again per attempt: kept across the attempts they were four registers too many.)
                    // At D > 64 only (RETRY): around the pass of the D <= 64 instantiations -- whose passes fit -- the loop cost 1.7 %.
                    int attempt = 0;
                    for (;;) {
                    const bool mine = attempt == 0 || (wave >> 1) == attempt - 1;   // (wave-uniform: this wave's planes are in)
                    // (the shapes, re-read per attempt: held from the top of the item across the passes they were spilled scalars)
                    const int H = RETRY ? KARG(int, a.H) : H_, W = RETRY ? KARG(int, a.W) : W_, C = RETRY ? KARG(int, a.C) : C_, D = RETRY ? KARG(int, a.D) : D_;
                    // ---- sample positions of this thread's planes of the pass (NO_CELL: no tap in the image, plane beyond D,
                    //      pixel beyond the image)
                    int cell[4];
                    float fw[4], fn[4];
                    {
                        ViewXform xf;
                        {
                            const v4f k0 = *reinterpret_cast<const v4f*>(&L.xf[v * 12]), k1 = *reinterpret_cast<const v4f*>(&L.xf[v * 12 + 4]),
                                      k2 = *reinterpret_cast<const v4f*>(&L.xf[v * 12 + 8]);
                            xf.kr[0] = k0.x; xf.kr[1] = k0.y; xf.kr[2] = k0.z; xf.kr[3] = k0.w; xf.kr[4] = k1.x; xf.kr[5] = k1.y;
                            xf.kr[6] = k1.z; xf.kr[7] = k1.w; xf.kr[8] = k2.x; xf.kt[0] = k2.y; xf.kt[1] = k2.z; xf.kt[2] = k2.w;
                            xf.separate = KARG(int, a.blas_mode);
                        }
                        const v4f c0 = *reinterpret_cast<const v4f*>(&L.cst[0]);
                        const v2f c1 = *reinterpret_cast<const v2f*>(&L.cst[4]);
                        const v4f dk = *reinterpret_cast<const v4f*>(&L.dcl[64 * h + 4 * tq]);
                        float t2a, t2b, t2c;
                        ray_term2(xf, ray[0], ray[1], ray[2], t2a, t2b, t2c);
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const int k = 64 * h + 4 * tq + j;
                            float ix, iy;
                            // (scalar instructions: packed fp32 -- two planes per v_pk_* instruction -- now and then loses the low
                            //  half of a result in lanes 48..63 beside v_mfma_f32_16x16x32_f16 on gfx950: wave_util.hpp)
                            plane_sample_pos_fast(xf, t2a, t2b, t2c, dk[j], c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, ix, iy);
                            cell[j] = cell_of(ix, iy, W, H, fw[j], fn[j]);
                            if (k >= D || !xlive || !mine) cell[j] = NO_CELL;
                            if (DIST_ABL & 1) {   // timing only: the position chain a second time (how much of the kernel is vector issue?)
                                float ix2, iy2, f2, g2;
                                plane_sample_pos_fast(xf, t2a, t2b, t2c, dk[j] * 1.0001f, c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, ix2, iy2);
                                const int c2 = cell_of(ix2, iy2, W, H, f2, g2);
                                asm volatile("" :: "v"(c2), "v"(f2), "v"(g2));
                            }
                        }
                        // (pinned: the optimiser otherwise carries the positions AND their floors to the combine instead of the fractions)
#pragma unroll
                        for (int j = 0; j < 4; ++j) { asm volatile("" : "+v"(fw[j])); asm volatile("" : "+v"(fn[j])); }
                    }
                    DSTAMP(2)   // (wait for the rays) sample positions
                    if (!centred) {
                        // (first pass of the trip) this thread's channel pairs of pixel n: -2 x' = -2 (r - mu) 2^e in one rounding
                        // (scaling by a power of two commutes with it), split into fp16 high and low parts (two channels per
                        // conversion), into the pixel-side operand image of the block; and its share of |r'|^2 and |r|^2.  Pair
                        // tq of round mm = channels 32 mm + 2 tq, + 1: chunk mm, K slots 2 tq, 2 tq + 1 (K slice tq >> 2 = wave);
                        // the tail round (wave 0, pair kq): K slots 2 kq (high part, against the texel's high part), 8 + 2 kq (high,
                        // against the texel's low part), 16 + 2 kq (low).  A feature beyond the fp16 range after scaling becomes
                        // inf - inf in the products: the pixel's costs are NaN (never a clamped number).
                        const float sc = L.cst[6], sm2 = -2.0f * sc;
                        float pr = 0.0f, pz = 0.0f;
                        int* const Bw = reinterpret_cast<int*>(L.Bs);   // (pairs of halves)
#pragma unroll
                        for (int mm = 0; mm < MP; ++mm) {
                            if (mm == NCHK && wave != 0) continue;   // uniform
                            const v2f mu2 = *reinterpret_cast<const v2f*>(&L.mus[32 * mm + 2 * tq]);
                            const float x0 = __builtin_fmaf(rv[mm][0], sm2, 2.0f * mu2.x), x1 = __builtin_fmaf(rv[mm][1], sm2, 2.0f * mu2.y);
                            const float s0 = rv[mm][0] * sc, s1 = rv[mm][1] * sc;
                            pr = __builtin_fmaf(x0, x0, pr); pr = __builtin_fmaf(x1, x1, pr);
                            pz = __builtin_fmaf(s0, s0, pz); pz = __builtin_fmaf(s1, s1, pz);
                            const h2 bh = __builtin_convertvector(v2f{x0, x1}, h2);
                            const h2 bl = __builtin_convertvector(v2f{x0 - (float)bh.x, x1 - (float)bh.y}, h2);
                            if (mm < NCHK) {
                                Bw[((mm * 4 + wave) * 16 + n) * 4 + kq] = __builtin_bit_cast(int, bh);
                                Bw[(((NCHK + mm) * 4 + wave) * 16 + n) * 4 + kq] = __builtin_bit_cast(int, bl);
                            } else {
                                Bw[(((2 * NCHK) * 4 + 0) * 16 + n) * 4 + kq] = __builtin_bit_cast(int, bh);
                                Bw[(((2 * NCHK) * 4 + 1) * 16 + n) * 4 + kq] = __builtin_bit_cast(int, bh);
                                Bw[(((2 * NCHK) * 4 + 2) * 16 + n) * 4 + kq] = __builtin_bit_cast(int, bl);
                            }
                        }
                        pr = 0.25f * pr;
                        pr = pr + xor16_f(pr); pr = pr + bperm_f(a32, pr);
                        pz = pz + xor16_f(pz); pz = pz + bperm_f(a32, pz);
                        if (kq == 0) *reinterpret_cast<v2f*>(&L.rp[(wave * 16 + n) * 2]) = v2f{pr, pz};
                    }
                    // ---- row table: the texels this thread's planes touch ---------------------------------------------------------
                    // Exact runs: consecutive planes whose cells share a row are folded first and go in as ONE update of that row (two
                    // LDS atomics, issued behind the last plane of the run) -- on a rectified pair all four planes of a thread, and
                    // all 256 threads, meet in one or two rows, and atomics on one address are served lane by lane.  Straight-line
                    // code: a plane continues the run of the plane before it, or closes it.
                    const int par = (state >> 1) & 1;
                    {
                        int ylo = 32767, yhi = -32768;
                        int rmin = 0, rmax = 0, ry = 0;
                        bool open = false;   // a run is open: row ry, texels rmin .. rmax
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const bool vj = cell[j] != NO_CELL;
                            const int yj = cell_y(cell[j]), xj = cell_x(cell[j]);   // (NO_CELL: -32768, -32768)
                            const bool cont = open && vj && yj == ry;
                            if (open && !cont) {   // the run ends in front of plane j
                                atomicMin(&L.ctab[par][(ry & 63) * 2], rmin);
                                atomicMax(&L.ctab[par][(ry & 63) * 2 + 1], rmax);
                            }
                            rmin = cont ? min(rmin, xj) : xj;
                            rmax = cont ? max(rmax, xj) : xj;
                            ry = yj;
                            open = vj;
                            ylo = min(ylo, vj ? yj : 32767);
                            yhi = max(yhi, yj);
                        }
                        if (open) {
                            atomicMin(&L.ctab[par][(ry & 63) * 2], rmin);
                            atomicMax(&L.ctab[par][(ry & 63) * 2 + 1], rmax);
                        }
                        const int wmin = wave_min_i(ylo), wmax = wave_max_i(yhi);
                        if (lane == 0 && wmin <= wmax) { atomicMin(&L.ired[par][0], wmin); atomicMax(&L.ired[par][1], wmax); }
                    }
                    DSTAMP(3)   // table atomics, (wait for the reference features) centring
                    PDEPTH_LDS_BARRIER();   // tables (and the operand images) complete
                    DSTAMP(4)   // barrier
                    if (!centred) {
#pragma unroll
                        for (int w = 0; w < 4; ++w) {
                            const v2f pp = *reinterpret_cast<const v2f*>(&L.rp[(w * 16 + n) * 2]);
                            rr = rr + pp.x; zc = zc + pp.y;
                        }
                        // the specials of the tail chunk (K slots 24..31 = lanes kq == 3): the constants that multiply the texel's
                        // pieces of N, and |r'|^2 as three fp16 pieces against the texel's constants (dist_layout.hpp) -- into the
                        // operand image (every wave writes the same values; a wave reads what it wrote itself).  |r'|^2 beyond
                        // the pieces' range, or not a number: NaN pieces, NaN costs.
                        if (kq == 3) {
                            const dist::Pieces pq = dist::split_pieces(rr < 2.0e9f ? rr : __builtin_nanf(""));
                            h8 sp;
                            sp[0] = (_Float16)dist::PIECE_C1; sp[1] = (_Float16)dist::PIECE_C2; sp[2] = (_Float16)dist::PIECE_C3;
                            sp[3] = pq.p1; sp[4] = pq.p2; sp[5] = pq.p3; sp[6] = (_Float16)0.f; sp[7] = (_Float16)0.f;
                            *reinterpret_cast<h8*>(&L.Bs[(((NAC - 1) * 4 + 3) * 16 + n) * 8]) = sp;
                        }
                        centred = true;
                    }
                    // the pixel-side operands of lane (n, kq): high[NCHK], low[NCHK], tail -- from LDS once per pass, or (DIST_BV_LDS)
                    // chunk by chunk in front of the multiplications of every block: twenty registers for a second texel operand set
                    h8 Bv[NAC];
                    const _Float16* const bsl = &L.Bs[(kq * 16 + n) * 8];
                    if (!DIST_BV_LDS) {
#pragma unroll
                        for (int i = 0; i < NAC; ++i) Bv[i] = *reinterpret_cast<const h8*>(bsl + i * 512);
                    }
                    // ---- the row table, cut into blocks of 16 texels: every wave for itself, lane = texel row yb + lane -----
                    const int yb = __builtin_amdgcn_readfirstlane(L.ired[par][0]), yt = __builtin_amdgcn_readfirstlane(L.ired[par][1]);
                    int nb = 0, lo = 0, nblk = 0, fb = 0;
                    bool fits = true;
                    if (yb <= yt) {
                        const int ncell = yt - yb + 1;
                        if (ncell + 1 > 64 || ncell + 1 > MAXB) {   // (every texel row takes a block; rows are kept modulo 64)
                            fits = false;
                        } else {
                            // the cells of rows lane - 1 and lane touch texel row lane (rows outside the pass: empty entries)
                            const v2i e0 = *reinterpret_cast<const v2i*>(&L.ctab[par][((yb + lane) & 63) * 2]);
                            const v2i e1 = *reinterpret_cast<const v2i*>(&L.ctab[par][((yb + lane - 1) & 63) * 2]);
                            const int l0 = min(e0.x, e1.x), h0 = max(e0.y, e1.y);
                            // (blocks start at a multiple of 4 texels of the padded row: whole 64-byte pieces for the loads of four lanes)
                            lo = ((l0 + dist::RING) & ~3) - dist::RING;
                            nblk = l0 <= h0 ? (h0 - lo + 2 + 15) >> 4 : 0;   // texels lo .. hi + 1
                            const int incl = wave_scan_incl(nblk);
                            nb = __builtin_amdgcn_readlane(incl, 63);
                            fits = nb <= MAXB;
                            fb = incl - nblk;
                        }
                    }
                    const int iflag = __builtin_amdgcn_readfirstlane(L.iflag);
                    if ((iflag & 3) != 0) fits = false;   // (the item is evaluated directly; an ill-conditioned one -- bit 2 -- on a packed source is not: nothing to gain)
                    if (DIST_FORCE_DIRECT == -1 || DIST_FORCE_DIRECT == v) fits = false;   // (test builds)
                    state ^= 2;
                    const int rowoff = 16 * fb - lo;    // slot of texel x of this lane's row = x + rowoff
                    const bool go = fits && nb > 0;
                    DSTAMP(5)   // operands from LDS, row table cut into blocks
                    // ---- Y = |s' - r'|^2 for the blocks of the pass, on the matrix pipe ------------------------------------
                    int sl0[4], sl1[4];   // slots of the top / bottom row of this thread's cells
                    {
                        // byte steps of the packed view (dist_layout.hpp: texel_offset): between the planes of a texel (PB), between
                        // the groups of four texels (GB), and what a lane adds for its texel n of a block.  A block = four whole
                        // groups: one load instruction (4 planes of 16 texels) reads four runs of 256 bytes, all the operands of
                        // a block lie within 5 KB.
                        constexpr int PB = dist::GROUP_PLANE_BYTES, GB = (8 * NCHK + 4) * PB;   // (= dist::group_bytes(C): nchk(C) = NCHK)
                        const int Wp = dist::wp(W);
                        const int ntex = (n / dist::GROUP) * GB + (n % dist::GROUP) * 16;
                        // (the view's address: formed here, per pass -- across the pass and its attempts it was three spilled scalars)
                        const char* srcv = RETRY ? KARG(const char*, packed) + (size_t)((size_t)b * KARG(int, a.V) + v) * dist::view_bytes(C, H, W) : srcv_v;
                        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)srcv, 0, (int)dist::view_bytes(C, H, W), 0x00020000);
                        const int voffA = opaque_v(ntex + kq * PB);
                        // a block's texel operands in consumption order: (high c, low c) for c < NCHK, tail; NS register sets: NS
                        // blocks of the wave are in flight, every chunk refilled with the operands of the block after the next
                        // right behind its last multiplication.  (One set -- round 5 -- left every block of a wave waiting a whole
                        // memory latency for its operands: three to four latencies per pass on a forward motion.)
                        h8 S[NS][NAC];
                        // Block j of the pass (lane j): the byte offset of its first texel in plane 0.  Its row = the lane rho with
                        // fb <= j < fb + nblk: the row lanes scatter their index through LDS (every wave writes the same values and
                        // reads back what it wrote), its first texel lo[rho] + 16 (j - fb[rho]).
                        int boff = 0;
                        if (go) {
#pragma unroll 1
                            for (int i = 0; i < nblk; ++i) L.brow[fb + i] = lane;
                            const int rho = L.brow[min(lane, MAXB - 1)];
                            const int xs = __builtin_amdgcn_ds_bpermute(4 * rho, lo) + 16 * (lane - __builtin_amdgcn_ds_bpermute(4 * rho, fb));
                            // (xs + RING is a multiple of 4: the block starts at a texel group)
                            boff = lane < nb ? ((yb + rho + dist::RING) * (Wp / dist::GROUP) + (xs + dist::RING) / dist::GROUP) * GB : OOB;
                        }
                        // (the tail chunk: planes high | low | high again | specials = the tail's planes 0, 1, 0, 2 for K slices 0 .. 3)
                        auto fetch = [&](int set, int i, int soff) {   // chunk i: planes 4 i .. 4 i + 3 (the lane's: + kq, in voffA)
                            if (DIST_ABL & 2) { S[set][i] = *reinterpret_cast<const h8*>(bsl + i * 512); return; }   // timing only: no texel loads
                            const int voff = i == NAC - 1 ? voffA - (kq == 2 ? 2 * PB : (kq == 3 ? PB : 0)) : voffA;
                            S[set][i] = __builtin_bit_cast(h8, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, soff + i * 4 * PB, 0));
                        };
                        auto fetch_block = [&](int set, int soff) {
#pragma unroll
                            for (int c = 0; c < NCHK; ++c) { fetch(set, c, soff); fetch(set, NCHK + c, soff); }
                            fetch(set, NAC - 1, soff);
                        };
                        // wave w: blocks w q .. w q + q - 1 (consecutive blocks: consecutive slots)
                        const int q = (nb + 3) >> 2, b0 = wave * q, b1 = min(nb, b0 + q);
                        if (DIST_XPRIO) __builtin_amdgcn_s_setprio(DIST_XPRIO);
                        if (go) {
#pragma unroll
                            for (int u = 0; u < NS; ++u)
                                if (b0 + u < b1) fetch_block(u, __builtin_amdgcn_readlane(boff, b0 + u));
                            // (the first blocks' operand loads go in front of the Q records', which are needed behind the pass's second barrier only:
                            // -0.8 % headline, -2.5 % config 5)
                            // the Q records of the pass's cells, from memory straight to LDS: blocks 4 g .. 4 g + 3 per instruction
                            // (lane = (block, texel)); wave w moves groups w and w + 4
#pragma unroll
                            for (int gq = 0; gq < (MAXB + 15) / 16; ++gq) {
                                const int g = wave + 4 * gq;
                                if (4 * g < nb) {   // uniform
                                    const int bo = __builtin_amdgcn_ds_bpermute(4 * (4 * g + kq), boff);
                                    const int vq = 4 * g + kq < nb ? bo + ntex : OOB;
                                    dma_b128(rsrc, lds_addr_of(&L.Qs[g * 256]), vq, QPL * PB);
                                }
                            }
                        }
                        // the slots of this thread's cells (under the first blocks' loads)
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const bool has = fits && cell[j] != NO_CELL;
                            const int r = has ? cell_y(cell[j]) - yb : 0;
                            const int o0 = __builtin_amdgcn_ds_bpermute(4 * r, rowoff), o1 = __builtin_amdgcn_ds_bpermute(4 * r + 4, rowoff);
                            const int cxx = cell_x(cell[j]);
                            sl0[j] = has ? cxx + o0 : -1;
                            sl1[j] = cxx + o1;
                        }
                        if (go) {
#pragma unroll 1
                            for (int bi = b0; bi < b1; bi += NS) {   // (rolled: unrolled, the compiler hoists the later blocks' work and spills)
#pragma unroll
                                for (int u = 0; u < NS; ++u) {
                                    const int bj = bi + u;
                                    if (bj >= b1) break;   // uniform
                                    const bool more = bj + NS < b1;   // uniform
                                    const int soff = more ? __builtin_amdgcn_readlane(boff, bj + NS) : 0;
                                    v4f acc = v4f{0.f, 0.f, 0.f, 0.f};
                                    if (DIST_BV_LDS) {
#pragma unroll
                                        for (int c = 0; c < NCHK; ++c) {
                                            const h8 bh = *reinterpret_cast<const h8*>(bsl + c * 512);
                                            const h8 bl = *reinterpret_cast<const h8*>(bsl + (NCHK + c) * 512);
                                            acc = DIST_MFMA(S[u][c], bh, acc);
                                            acc = DIST_MFMA(S[u][NCHK + c], bh, acc);
                                            if (more) fetch(u, NCHK + c, soff);
                                            acc = DIST_MFMA(S[u][c], bl, acc);
                                            if (more) fetch(u, c, soff);
                                        }
                                        const h8 bt = *reinterpret_cast<const h8*>(bsl + (NAC - 1) * 512);
                                        acc = DIST_MFMA(S[u][NAC - 1], bt, acc);
                                        if (more) fetch(u, NAC - 1, soff);
                                    } else {
#pragma unroll
                                        for (int c = 0; c < NCHK; ++c) {
                                            acc = DIST_MFMA(S[u][c], Bv[c], acc);
                                            acc = DIST_MFMA(S[u][c], Bv[NCHK + c], acc);
                                            if (more) fetch(u, c, soff);
                                            acc = DIST_MFMA(S[u][NCHK + c], Bv[c], acc);
                                            if (more) fetch(u, NCHK + c, soff);
                                        }
                                        acc = DIST_MFMA(S[u][NAC - 1], Bv[NAC - 1], acc);
                                        if (more) fetch(u, NAC - 1, soff);
                                    }
                                    // Y[texel 4 kq ..][pixel n] of the block
                                    *reinterpret_cast<v4f*>(&L.Ys[n * XSTRIDE + (n >> 3) * DIST_YSKEW + 16 * bj + 4 * kq]) = acc;
                                }
                            }
                            DSTAMP(6)   // slots, loads + multiplications
                            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's Q records have landed in LDS
                        }
                        if (DIST_XPRIO) __builtin_amdgcn_s_setprio(0);
                    }
                    DSTAMP(7)   // wait for the Q records
                    PDEPTH_LDS_BARRIER();   // Y and the Q records of the pass are complete
                    DSTAMP(8)   // barrier
                    // (every wave has read the tables of this pass: wave 1 cleans them for the pass after the next)
                    if (wave == 1) {
                        *reinterpret_cast<v2i*>(&L.ctab[par][lane * 2]) = v2i{INT_MAX, INT_MIN};
                        if (lane == 0) { L.ired[par][0] = INT_MAX; L.ired[par][1] = INT_MIN; }
                    }

                    // ---- combine: cost of this thread's planes of the pass ---------------------------------------------
                    if (!fits) {   // (a half: evaluated directly behind the view loop; the whole pass: again, in halves)
                        if (attempt != 0) failmask |= 1u << ((v * NH + h) * 2 + attempt - 1);
                    } else if (attempt == 0 || (wave >> 1) == attempt - 1) {   // (`mine`, formed again: kept from the top of the attempt it is a lane mask in two scalar registers)
                        const float cinv = L.cst[7];
                        const float* yr = &L.Ys[n * XSTRIDE + (n >> 3) * DIST_YSKEW];
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            // (no tap inside the image: the taps read zero, cost = |r|^2 -- and NaN where the position itself
                            //  is not finite, as the reference's weights inf - floor(inf) make it)
                            float qv = zc + (fw[j] + fn[j]) * 0.0f;
                            if (sl0[j] >= 0) {
                                const int s0 = sl0[j], s1 = sl1[j];
                                const float Y00 = yr[s0], Y01 = yr[s0 + 1], Y10 = yr[s1], Y11 = yr[s1 + 1];
                                const v4f qa = *reinterpret_cast<const v4f*>(&L.Qs[s0 * 4]);   // Dx0, Dy0, Dd, Dx1
                                const float dy1 = L.Qs[s0 * 4 + 5];                            // Dy0 of the right neighbour
                                const float ex = 1.0f - fw[j], ey = 1.0f - fn[j];
                                const float w00 = ey * ex, w01 = ey * fw[j], w10 = fn[j] * ex, w11 = fn[j] * fw[j];
                                const float ys = __builtin_fmaf(w11, Y11, __builtin_fmaf(w10, Y10, __builtin_fmaf(w01, Y01, w00 * Y00)));
                                const float qi = __builtin_fmaf(w11, qa.z, __builtin_fmaf(w10, qa.y, w01 * qa.x));
                                const float qo = __builtin_fmaf(w01, dy1, w10 * qa.w);
                                qv = ys - __builtin_fmaf(w11, qo, w00 * qi);
                            }
                            cost[4 * h + j] = cost[4 * h + j] + qv * cinv;
                        }
                    }
                    DSTAMP(9)   // combine
                    if (!RETRY) {   // (D <= 64: straight-line, a pass that does not fit goes to the direct evaluation whole)
                        if (!fits) failmask |= 3u << ((v * NH + h) * 2);
                        break;
                    }
                    if (attempt == 0) {
                        if (fits) break;
                        if ((iflag & 3) != 0 || DIST_FORCE_DIRECT == -1 || DIST_FORCE_DIRECT == v) {   // (no half of this item fits either)
                            failmask |= 3u << ((v * NH + h) * 2);
                            break;
                        }
                    } else if (attempt == 2) {
                        break;
                    }
                    ++attempt;
                    }   // attempts
                }
            }

            if (failmask != 0) {   // uniform
                // Half-passes left to the direct evaluation: the reference's own form on the packed features -- per plane the four
                // taps of every group of 8 channels by two 16-byte loads each (high + low parts), the pixel's centred reference
                // features from the operand image in LDS.  An item whose features did not fit the fp16 range: NaN.
                // (the shapes re-read here: kept from the top of the item for this rarely run block, they were spilled scalars)
                const int H = KARG(int, a.H), W = KARG(int, a.W), V = KARG(int, a.V), C = KARG(int, a.C), D = KARG(int, a.D);
                const int b = opaque_s((state >> 2) - 1);   // (the batch item in work)
                const v4f c0 = *reinterpret_cast<const v4f*>(&L.cst[0]), c1 = *reinterpret_cast<const v4f*>(&L.cst[4]);
                const bool ovf = (L.iflag & 1) != 0;
                (void)0;
#pragma unroll 1
                for (int vhh = 0; vhh < V * NH * 2; ++vhh) {
                    if (!(failmask >> vhh & 1u)) continue;
                    if (tid == 0) ++n_direct;
                    if ((wave >> 1) != (vhh & 1)) continue;   // (the planes of the other pair of waves; no barrier in this loop)
                    const int vh = vhh >> 1, v = vh / NH, h = vh - v * NH;
                    ViewXform xf;
#pragma unroll
                    for (int i = 0; i < 9; ++i) xf.kr[i] = L.xf[v * 12 + i];
#pragma unroll
                    for (int i = 0; i < 3; ++i) xf.kt[i] = L.xf[v * 12 + 9 + i];
                    xf.separate = KARG(int, a.blas_mode);
                    float t2a, t2b, t2c;
                    ray_term2(xf, ray[0], ray[1], ray[2], t2a, t2b, t2c);
                    const char* srcv = KARG(const char*, packed) + (size_t)((size_t)b * V + v) * dist::view_bytes(C, H, W);
#pragma unroll 1
                    for (int j = 0; j < 4; ++j) {
                        const int k = 64 * h + 4 * tq + j;
                        float ix, iy, fwj, fnj;
                        plane_sample_pos_fast(xf, t2a, t2b, t2c, L.dcl[k], c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, ix, iy);
                        int cellj = cell_of(ix, iy, W, H, fwj, fnj);
                        if (k >= D || !xlive) cellj = NO_CELL;
                        float qv = zc + (fwj + fnj) * 0.0f;   // (no tap inside the image)
                        if (cellj != NO_CELL) {
                            const int cxx = cell_x(cellj), cyy = cell_y(cellj);
                            const float ex = 1.0f - fwj, ey = 1.0f - fnj;
                            const float w00 = ey * ex, w01 = ey * fwj, w10 = fnj * ex, w11 = fnj * fwj;
                            // (the four taps' texels: neighbours in a row may lie in two texel groups)
                            const long long t00 = dist::texel_offset(C, H, W, 0, cyy + dist::RING, cxx + dist::RING), t01 = dist::texel_offset(C, H, W, 0, cyy + dist::RING, cxx + dist::RING + 1);
                            const long long t10 = dist::texel_offset(C, H, W, 0, cyy + dist::RING + 1, cxx + dist::RING), t11 = dist::texel_offset(C, H, W, 0, cyy + dist::RING + 1, cxx + dist::RING + 1);
                            const long long pstep = dist::texel_offset(C, H, W, 1, 0, 0);   // from a plane to the next
                            float part = 0.0f;
#pragma unroll DIST_DIRECT_UNROLL
                            for (int g = 0; g < 4 * NCHK + 1; ++g) {
                                // planes of the group's high and low parts; the pixel's: chunk g >> 2 (tail: 2 NCHK), K slice g & 3 (tail: 0 | 2)
                                const bool tail = g == 4 * NCHK;
                                const char* ph = srcv + (tail ? 8 * NCHK : g) * pstep;
                                const char* pl = srcv + (tail ? 8 * NCHK + 1 : 4 * NCHK + g) * pstep;
                                const h8 a00 = *reinterpret_cast<const h8*>(ph + t00), a01 = *reinterpret_cast<const h8*>(ph + t01);
                                const h8 a10 = *reinterpret_cast<const h8*>(ph + t10), a11 = *reinterpret_cast<const h8*>(ph + t11);
                                const h8 l00 = *reinterpret_cast<const h8*>(pl + t00), l01 = *reinterpret_cast<const h8*>(pl + t01);
                                const h8 l10 = *reinterpret_cast<const h8*>(pl + t10), l11 = *reinterpret_cast<const h8*>(pl + t11);
                                const h8 rh = *reinterpret_cast<const h8*>(&L.Bs[(((tail ? 2 * NCHK : (g >> 2)) * 4 + (tail ? 0 : (g & 3))) * 16 + n) * 8]);
                                const h8 rl = *reinterpret_cast<const h8*>(&L.Bs[(((tail ? 2 * NCHK : NCHK + (g >> 2)) * 4 + (tail ? 2 : (g & 3))) * 16 + n) * 8]);
#pragma unroll
                                for (int i = 0; i < 8; ++i) {
                                    float val = ((float)a00[i] + (float)l00[i]) * w00;
                                    val = __builtin_fmaf((float)a01[i] + (float)l01[i], w01, val);
                                    val = __builtin_fmaf((float)a10[i] + (float)l10[i], w10, val);
                                    val = __builtin_fmaf((float)a11[i] + (float)l11[i], w11, val);
                                    const float rq = -0.5f * ((float)rh[i] + (float)rl[i]);
                                    const float diff = val - rq;
                                    part = __builtin_fmaf(diff, diff, part);
                                }
                            }
                            qv = part + (fwj + fnj) * 0.0f;
                        }
                        if (ovf) qv = __builtin_nanf("");
                        const float cj = qv * L.cst[7];
#pragma unroll
                        for (int jj = 0; jj < NC; ++jj) cost[jj] = cost[jj] + (jj == 4 * h + j ? cj : 0.0f);
                    }
                }
            }
            // ---- epilogue: cost store, log-softmax over D, expectation ----------------------------------------------------
            // (buffer stores: one 32-bit lane offset, the plane as the scalar offset: plane 64 h + 16 wave + 4 kq + j)
            // (the planes' scalar offsets are formed here, from a value the optimiser cannot trace back: hoisted to the top of
            //  the item they were four spilled scalars, read back lane by lane with five wait states each in front of a store)
            resolve_next();
            const int HW4 = KARG(int, a.H) * KARG(int, a.W) * 4;   // (re-read: HW * 4 kept from the top of the block was a spilled scalar)
            const int ovoff = xlive ? 4 * kq * HW4 + p * 4 : OOB;
            const int pl0 = 16 * wave * HW4;
            if (float* const cost_out = KARG(float*, a.cost_out)) {
                const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc((void*)(cost_out + (size_t)b * D * (HW4 >> 2)), 0, D * HW4, 0x00020000);
#pragma unroll
                for (int j = 0; j < NC; ++j)   // (planes beyond D lie beyond the descriptor: dropped)
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, cost[j]), rc, ovoff, pl0 + (64 * (j >> 2) + (j & 3)) * HW4, DIST_STORE_AUX);
            }
            float* const logp_out = KARG(float*, a.logp_out);
            float* const depth_out = KARG(float*, a.depth_out);
            if (logp_out || depth_out) {
                // per wave: max, sum exp, sum d exp over its planes of pixel n; merged over the four waves through LDS
                {
                    float mx = -INFINITY;
#pragma unroll
                    for (int j = 0; j < NC; ++j)
                        mx = max_raw(mx, 64 * (j >> 2) + 4 * tq + (j & 3) < D ? cost[j] : -INFINITY);
                    mx = max_raw(mx, xor16_f(mx));
                    mx = max_raw(mx, bperm_f(a32, mx));
                    float ssum = 0.0f, esum = 0.0f;
#pragma unroll
                    for (int j = 0; j < NC; ++j) {
                        const int k = 64 * (j >> 2) + 4 * tq + (j & 3);
                        const float ek = k < D ? dist_exp(cost[j] - mx) : 0.0f;
                        ssum = ssum + ek;
                        esum = __builtin_fmaf(L.dcl[k], ek, esum);
                    }
                    ssum = ssum + xor16_f(ssum); ssum = ssum + bperm_f(a32, ssum);
                    esum = esum + xor16_f(esum); esum = esum + bperm_f(a32, esum);
                    if (kq == 0) *reinterpret_cast<v4f*>(&L.red[(wave * 16 + n) * 4]) = v4f{mx, ssum, esum, 0.0f};
                }
                DSTAMP(10)   // cost stores, partial softmax
                publish_next();
                if (!DIST_ABL_NOB3) PDEPTH_LDS_BARRIER();
                {
                    v4f part[4];
#pragma unroll
                    for (int w = 0; w < 4; ++w) part[w] = *reinterpret_cast<const v4f*>(&L.red[(w * 16 + n) * 4]);
                    const float M = max_raw(max_raw(part[0].x, part[1].x), max_raw(part[2].x, part[3].x));
                    float S_ = 0.0f, E = 0.0f;
#pragma unroll
                    for (int w = 0; w < 4; ++w) {
                        // (a wave whose planes all lie beyond D: max -inf, sums 0)
                        const float scw = part[w].x == -INFINITY ? 0.0f : dist_exp(part[w].x - M);
                        S_ = __builtin_fmaf(part[w].y, scw, S_);
                        E = __builtin_fmaf(part[w].z, scw, E);
                    }
                    const float ls = logf(S_);
                    if (logp_out) {
                        const __amdgpu_buffer_rsrc_t rl = __builtin_amdgcn_make_buffer_rsrc((void*)(logp_out + (size_t)b * D * (HW4 >> 2)), 0, D * HW4, 0x00020000);
#pragma unroll
                        for (int j = 0; j < NC; ++j)
                            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, (cost[j] - M) - ls), rl, ovoff,
                                                                  pl0 + (64 * (j >> 2) + (j & 3)) * HW4, DIST_STORE_AUX);
                    }
                    if (depth_out && xlive && tq == 0) depth_out[(size_t)b * (HW4 >> 2) + p] = E / S_;
                }
            } else {
                publish_next();
                PDEPTH_LDS_BARRIER();   // (every wave is done with the block's operand image before the next block's centring)
            }
            DSTAMP(11)   // barrier + merge + stores
        }   // the pixel block
    }   // items
#ifdef DIST_STAMPS
    if ((threadIdx.x & 63) == 0)
        for (int i = 0; i < 12; ++i) atomicAdd(reinterpret_cast<unsigned long long*>(KARG(int*, queue) + 8) + i, stamp_acc[i]);
#endif

    // Leaving.  With a queue: the last workgroup to leave zeroes the queue counters (the next call on this workspace needs no
    // clearing launch) and publishes the call's count of directly evaluated pixel blocks.  Without one (a workgroup per item)
    // nothing was counted that needs resetting, and a returning atomic per workgroup on one address is what such a launch
    // cannot afford: a workgroup with direct passes -- rare -- adds them to the published count itself, tagged with the call's
    // nonce (a count left by another call is replaced).
    if (opaque_v((int)threadIdx.x) == 0) {
        int* queue = KARG(int*, queue);
        const int nonce = KARG(int, nonce);
        if (ONE_EACH) {
            if (n_direct) {
                int old = atomicAdd(&queue[DIST_DIRECT_LAST_SLOT], 0);
                for (;;) {
                    const int nv = (old >> 20) == nonce ? old + n_direct : (nonce << 20) | n_direct;
                    const int seen = atomicCAS(&queue[DIST_DIRECT_LAST_SLOT], old, nv);
                    if (seen == old) break;
                    old = seen;
                }
            }
        } else {
            if (n_direct) atomicAdd(&queue[DIST_DIRECT_SLOT], n_direct);
            const int done = atomicAdd(&queue[DIST_DONE_SLOT], 1);
            if (done == KARG(int, nblk) - 1) {
                const int nd = atomicAdd(&queue[DIST_DIRECT_SLOT], 0);
                for (int q = 0; q < 8; ++q) KARG(int*, qcnt)[q * KARG(int, qstride)] = 0;
                queue[DIST_DONE_SLOT] = 0;
                queue[DIST_DIRECT_SLOT] = 0;
                queue[DIST_DIRECT_LAST_SLOT] = (nonce << 20) | nd;   // diagnostics: pixel blocks of this call evaluated directly
            }
        }
    }
}

// 1 .. 2047, another one per launch of any instantiation (kernels.hpp: DIST_NONCE_SLOT)
inline int next_nonce() {
    static std::atomic<unsigned> launches{0};
    return (int)(launches.fetch_add(1) % 2047u) + 1;
}

template <int NCHK, int NH>
hipError_t launch_inst(const SweepArgs& a, const char* packed, const float* stats, int* queue, int tiles_x, int tiles, hipStream_t stream) {
    auto kern = sweep_dist_kernel<NCHK, NH>;
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
    DistArgs da;
    da.a = a; da.packed = packed; da.stats = stats; da.queue = queue; da.tiles_x = tiles_x; da.ntile = tiles;
    // The queue counters: memory-side atomics on ONE line are served one after the other (~13 ns each: 32 768 pops of the
    // headline launch on eight neighbouring ints took 0.42 ms whatever the kernel did in between); 256 bytes apart they are not.
    {
        const int nflags = (int)(sweep_ws_flag_only_bytes(a.B, a.H, a.W) / sizeof(int));   // >= 64
        int qs = DIST_QSTRIDE;
        while (qs * 8 > nflags) qs >>= 1;
        da.qcnt = queue - nflags;
        da.qstride = qs;
    }
    da.nonce = next_nonce();
    // up to DIST_ONE_EACH_X items per resident workgroup there is no queue: a workgroup per item (sweep_dist_knobs.hpp)
    const long long need = 8ll * ((tiles + 7) / 8) * 4 * a.B;   // a workgroup per pixel block of the largest XCD band, times 8
    da.one_each = need <= DIST_ONE_EACH_X * nblk;
    if (da.one_each) nblk = need;
    da.nblk = (int)nblk;
    hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(256), 0, stream, da);
    return hipGetLastError();
}

}  // namespace

// shapes the kernel is built for: L2, D <= 128, C <= 72, at most 8 source views; a view's planes within 2^31 bytes
bool sweep_dist_supports(const SweepArgs& a) {
    const long long hw = (long long)a.H * a.W;
    return a.metric == 0 && a.D <= 128 && a.C <= dist::MAX_C && a.V <= DIST_MAXV && a.W <= 32760 && a.H <= 32760 &&
           hw * a.D * 4 < (1ll << 31) && hw * a.C * 4 < (1ll << 31) && dist::view_bytes(a.C, a.H, a.W) < (1ll << 31) && hw * 12 < (1ll << 31);
}

// NCHW entry: channel statistics + pack kernel (launch_pack_dist), then the sweep kernel.  Packed entry: the sweep kernel alone.
hipError_t launch_sweep_dist(const SweepArgs& a, void* workspace, hipStream_t stream, bool packed_ready) {
    const int tiles_x = (a.W + 15) / 16, tiles_y = (a.H + 3) / 4, tiles = tiles_x * tiles_y;
    const char* packed = static_cast<const char*>(workspace) + sweep_ws_flag_bytes(a.B, a.H, a.W);
    int* queue = reinterpret_cast<int*>(static_cast<char*>(workspace) + sweep_ws_flag_only_bytes(a.B, a.H, a.W));
    const float* stats = reinterpret_cast<const float*>(static_cast<char*>(workspace) + sweep_ws_stats_offset(a.B, a.V, a.C, a.H, a.W));
    if (!packed_ready) {
        hipError_t e = launch_pack_dist(a, workspace, stream, /*fuse_stats=*/true);
        if (e != hipSuccess) return e;
    }
    const int nck = dist::nchk(a.C);
    hipError_t e;
    if (a.D <= 64)
        e = nck == 0 ? launch_inst<0, 1>(a, packed, stats, queue, tiles_x, tiles, stream)
                     : (nck == 1 ? launch_inst<1, 1>(a, packed, stats, queue, tiles_x, tiles, stream) : launch_inst<2, 1>(a, packed, stats, queue, tiles_x, tiles, stream));
    else
        e = nck == 0 ? launch_inst<0, 2>(a, packed, stats, queue, tiles_x, tiles, stream)
                     : (nck == 1 ? launch_inst<1, 2>(a, packed, stats, queue, tiles_x, tiles, stream) : launch_inst<2, 2>(a, packed, stats, queue, tiles_x, tiles, stream));
    if (e != hipSuccess || a.src == nullptr) return e;
    // NCHW entry: the items the kernel left alone (fp16 overflow, guard, conditioning: flag 1 of their statistics row) are
    // evaluated by the gather kernel on the caller's tensor, in the reference's own rounding; on the usual input every block
    // of this launch reads one flag and leaves
    return launch_sweep_direct_items(a, reinterpret_cast<const int*>(stats + STATS_FLAGS) + 1, STATS_STRIDE, stream);
}

}  // namespace pdepth
