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

#include "geometry.hpp"
#include "kernels.hpp"

namespace pdepth {

namespace {

constexpr int TW = 16, TH = 4;  // tile of reference pixels per block; wave w owns row w
constexpr int NW = 4;           // waves per block
constexpr int NT = 64 * NW;
constexpr int NS = 32;          // cell slots per pixel and pass
constexpr int JS = 16;          // plane steps per lane and window: a window is 4 * JS = 64 planes
constexpr int WT = 1024;        // window texels per ring buffer
constexpr int SLOTS = WT / NT;  // DMA instructions per wave, chunk and buffer (64 texels each)
constexpr int BUF_BYTES = WT * 16;
constexpr int RING_BYTES = 2 * BUF_BYTES;        // two window buffers; aliased by the X dumps [NW][NS][16 px][4]
constexpr int DUMP_WAVE_BYTES = NS * 256;
constexpr int GRAMA_OFF = RING_BYTES;            // [WT] float4 (N, H, V, D1)
constexpr int GRAMB_OFF = GRAMA_OFF + WT * 16;   // [WT] float   D2
constexpr int DTAB_OFF = GRAMB_OFF + WT * 4;     // [D] depth candidates
static_assert(NW * DUMP_WAVE_BYTES <= RING_BYTES, "X dumps must fit the window ring");
static_assert(NW * 16 * NS * 4 <= BUF_BYTES, "cell lists must fit ring buffer 1");
static_assert(DTAB_OFF + 1024 <= 65536, "packed 16-bit LDS addresses");

typedef int v4i __attribute__((ext_vector_type(4)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(3))) v4f* lds_v4f;
typedef const __attribute__((address_space(3))) v4i* lds_v4i;
typedef const __attribute__((address_space(3))) float* lds_f;
typedef __attribute__((address_space(3))) float* lds_fw;
typedef __attribute__((address_space(3))) int* lds_iw;

// ---- DPP helpers ---------------------------------------------------------------------------------------
#define CELLS_DPP_I(v, ctrl) __builtin_amdgcn_update_dpp((v), (v), (ctrl), 0xf, 0xf, false)
#define CELLS_DPP_F(v, ctrl) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, (v)), __builtin_bit_cast(int, (v)), (ctrl), 0xf, 0xf, false))
constexpr int QP_XOR1 = 0xB1, QP_XOR2 = 0x4E;  // quad_perm [1,0,3,2], [2,3,0,1]
constexpr int QP_SHR1 = 0x90, QP_SHR2 = 0x44;  // quad_perm [0,0,1,2], [0,1,0,1]
constexpr int QP_B0 = 0x00, QP_B1 = 0x55, QP_B2 = 0xAA, QP_B3 = 0xFF;  // broadcast lane c of the quad

// Wave-wide min / max with a scalar result (all 64 lanes active): four DPP steps, rows combined on the scalar unit.
#define CELLS_STEP(OP, ctrl) v = OP(v, __builtin_amdgcn_update_dpp(v, v, ctrl, 0xf, 0xf, false))
__device__ __forceinline__ int wave_min_s(int v) {
    CELLS_STEP(min, 0xB1); CELLS_STEP(min, 0x4E); CELLS_STEP(min, 0x141); CELLS_STEP(min, 0x140);
    return min(min(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)),
               min(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}
__device__ __forceinline__ int wave_max_s(int v) {
    CELLS_STEP(max, 0xB1); CELLS_STEP(max, 0x4E); CELLS_STEP(max, 0x141); CELLS_STEP(max, 0x140);
    return max(max(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)),
               max(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}
#undef CELLS_STEP

// ---- LDS-DMA and hand-counted waits (the compiler must not know these loads write LDS; see sweep_tiled.hip) ----
__device__ __forceinline__ void dma_b128(v4i rsrc, unsigned lds_addr, int voff, int soff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                 :: "s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff) : "memory", "m0");
}
__device__ __forceinline__ void dma_b32(v4i rsrc, unsigned lds_addr, int voff, int soff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dword %1, %2, %3 offen lds"
                 :: "s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff) : "memory", "m0");
}
// Everything this wave has issued to vector memory has completed: its LDS-DMA of the chunk, and (known to the
// compiler, which adds its own wait in front of the first use) the reference feature loaded beside it.
__device__ __forceinline__ void wait_all() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ void lds_wait() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
__device__ __forceinline__ v4i make_rsrc(const void* base, int bytes) {
    const unsigned long long p = reinterpret_cast<unsigned long long>(base);
    v4i r;
    r.x = (int)(unsigned)p; r.y = (int)(unsigned)(p >> 32) & 0xffff; r.z = bytes; r.w = 0x00020000;
    return r;
}
__device__ __forceinline__ unsigned lds_addr_of(const void* p) {
    return (unsigned)(unsigned long long)(__attribute__((address_space(3))) const void*)p;
}

// Values the optimiser must re-derive where they are used: without this it hoists dozens of per-plane invariants
// (addresses, masks, products of kernel arguments) to the top of the kernel and spills them.
__device__ __forceinline__ int opaque_v(int x) { asm volatile("" : "+v"(x)); return x; }
__device__ __forceinline__ int opaque_s(int x) { asm volatile("" : "+s"(x)); return x; }

// A kernel argument used once or twice per tile: re-read from the kernarg segment at the point of use instead of
// occupying scalar registers for the whole kernel (`a` is the first kernel argument).
template <typename T>
__device__ __forceinline__ T cold_arg(size_t offset) {
    typedef const char __attribute__((address_space(4))) * kptr;
    typedef const volatile T __attribute__((address_space(4))) * vptr;
    return *(vptr)((kptr)__builtin_amdgcn_kernarg_segment_ptr() + offset);
}
#define CELLS_ARG(type, field) cold_arg<type>(offsetof(SweepArgs, field))

constexpr int KEY_NONE = INT_MIN;  // cell key of a plane without any tap inside the image

}  // namespace

// NWIN = number of 64-plane windows (1: D <= 64, 2: D <= 128); the costs of a lane live in NWIN * 16 registers.
template <int NWIN>
__global__ __launch_bounds__(NT, 3) void sweep_cells_kernel(SweepArgs a, const float4* __restrict__ packed,
                                                            int* __restrict__ tile_flags, int* __restrict__ queue,
                                                            int tiles_x, int ntile) {
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
    // this wave's cell lists [16 pixels][NS]: in ring buffer 1, which is dead between the block-wide exchange of a
    // pass (every wave is past its previous plane loop) and the barrier of chunk 0 (after which chunk 1 is staged into it)
    const unsigned clist0 = lds0 + BUF_BYTES + wave * (16 * NS * 4);

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
    const bool queued = (int)gridDim.x < 8 * ((ntile + 7) / 8) * a.B;
    if (queued) {
        if (tid == 0) s_item[0] = atomicAdd(&queue[xcd], 1);
        __syncthreads();
    }
    int item = queued ? s_item[0] : (int)(blockIdx.x >> 3), item_par = 0;
    while (item < nitems) {
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

        float cost[NWIN * JS];  // cost of plane 64 m + 4 j + g at [m * JS + j]
#pragma unroll
        for (int i = 0; i < NWIN * JS; ++i) cost[i] = 0.0f;
        bool bail = false;  // block-uniform: the tile goes to the gather kernel

        for (int v = 0; v < a.V && !bail; ++v) {
            ViewXform xf;
            make_view_xform(CELLS_ARG(const float*, K) + b * 9, CELLS_ARG(const float*, R) + ((size_t)b * a.V + v) * 9,
                            CELLS_ARG(const float*, t) + ((size_t)b * a.V + v) * 3, CELLS_ARG(int, blas_mode), xf);
            float t2a, t2b, t2c;
            ray_term2(xf, r0, r1, r2, t2a, t2b, t2c);
            const float4* srcv = packed + ((size_t)b * a.V + v) * (nchunk + 2) * HW;
            const v4i src_rsrc = make_rsrc(srcv, (nchunk + 2) * HW * 16);
            const __amdgpu_buffer_rsrc_t ref_buf = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<float*>(CELLS_ARG(const float*, ref) + (size_t)b * CELLS_ARG(long long, ref_bstride)), (short)0, a.C * HW * 4, 0x00020000);

#pragma unroll
            for (int m = 0; m < NWIN; ++m) {
                const int kw = m * 4 * JS;       // first plane of the window
                if (kw >= a.D || bail) break;    // uniform
                const int nstep = min(JS, (a.D - kw + 3) >> 2);  // plane steps of this window

                // ---- positions of this lane's planes of the window -----------------------------------
                float fx[JS], fy[JS];  // fractional sample position inside the cell
                int ki[JS];            // cell key (y0 << 16 | x0 & 0xffff) or KEY_NONE; later the packed plane info
                const int gk = opaque_v(kw + g);
#pragma unroll
                for (int j = 0; j < JS; ++j) {
                    const int k = gk + 4 * j;
                    float ix, iy;
                    plane_sample_pos_fast(xf, t2a, t2b, t2c, dtab[min(k, a.D - 1)], cx, cy, rcx, rcy, half_w, half_h, ix, iy);
                    const float xfl = floorf(ix), yfl = floorf(iy);
                    fx[j] = ix - xfl;
                    fy[j] = iy - yfl;
                    const int x0 = (int)fminf(fmaxf(xfl, -2.0f), (float)(a.W + 1));
                    const int y0 = (int)fminf(fmaxf(yfl, -2.0f), (float)(a.H + 1));
                    const bool any = live & (k < a.D) & (ix == ix) & (iy == iy) & ((unsigned)(x0 + 1) < (unsigned)(a.W + 1)) &
                                     ((unsigned)(y0 + 1) < (unsigned)(a.H + 1));
                    ki[j] = any ? ((y0 << 16) | (x0 & 0xffff)) : KEY_NONE;
                    if (j & 1) __builtin_amdgcn_sched_barrier(0);  // two divide chains at a time (register pressure)
                }

                int jlo = 0;
                while (jlo < nstep) {  // passes of this window (block-uniform)
                    // ---- scan: cell slot of every plane from step jlo on, bounding box, cut point ----
                    int slot[JS];
                    unsigned newmask = 0;   // bit j: the plane of step j opens a new cell
                    int run = 0, jok = jlo, carry = KEY_NONE;
                    int bx0 = INT_MAX, by0 = INT_MAX, bx1 = INT_MIN, by1 = INT_MIN;
#pragma unroll
                    for (int j = 0; j < JS; ++j) {
                        slot[j] = 0;
                        if (j >= jlo && j < nstep) {  // uniform
                            const int key = ki[j];
                            const int left = CELLS_DPP_I(key, QP_SHR1);
                            const int prev = g == 0 ? carry : left;
                            const int isn = (key != KEY_NONE && key != prev) ? 1 : 0;
                            int s1 = CELLS_DPP_I(isn, QP_SHR1);
                            s1 = isn + (g >= 1 ? s1 : 0);
                            int s2 = CELLS_DPP_I(s1, QP_SHR2);
                            s2 = s1 + (g >= 2 ? s2 : 0);
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
                            if (j >= jlo && j < jhi_ && ki[j] != KEY_NONE) {
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
                    if (bail) break;

                    // ---- cell list -> this lane's corner addresses; plane info -------------------------
                    int nc = 0;
#pragma unroll
                    for (int j = 0; j < JS; ++j) {
                        if (j >= jlo && j < jhi) {  // uniform
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
                    auto stage = [&](int bufi, int ch) {
#pragma unroll
                        for (int sl = 0; sl < SLOTS; ++sl)
                            if (sl * NT + wave * 64 < wtex)  // wave-uniform
                                dma_b128(src_rsrc, my_lds + bufi * BUF_BYTES + sl * NT * 16, so[sl], ch * HW * 16);
                    };
                    const int ro = p * 4;
                    // reference feature (channel 4 ch + g) of the chunk in flight: a load the compiler tracks (an asm
                    // load's destination register may be copied before the data has landed)
                    float rnx;
                    auto fetch_ref = [&](int ch) {
                        const int c = ch * 4 + g;  // beyond C: 0, like the packed source
                        rnx = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ref_buf, c < a.C ? ro + c * HW * 4 : 0x7fffffff, 0, 0));
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
                    fetch_ref(0);
                    stage(0, 0);

                    float X[NS];
#pragma unroll
                    for (int i = 0; i < NS; ++i) X[i] = 0.0f;
                    float rr = 0.0f;

                    // ---- channel loop ---------------------------------------------------------------------
#define CELLS_CELL(i_, BUF)                                                                               \
    {                                                                                                     \
        const unsigned ad = ((i_) & 1) ? ((unsigned)addrp[(i_) >> 1] >> 16) : ((unsigned)addrp[(i_) >> 1] & 0xffffu); \
        const v4f t = *(lds_v4f)(size_t)(ad + (BUF) * BUF_BYTES);                                         \
        float x_ = X[i_];                                                                                 \
        x_ = __builtin_fmaf(t.x, rb0, x_); x_ = __builtin_fmaf(t.y, rb1, x_);                             \
        x_ = __builtin_fmaf(t.z, rb2, x_); x_ = __builtin_fmaf(t.w, rb3, x_);                             \
        X[i_] = x_;                                                                                       \
    }
#define CELLS_CHUNK(BUF)                                                                                  \
    {                                                                                                     \
        const float rb0 = CELLS_DPP_F(rc, QP_B0), rb1 = CELLS_DPP_F(rc, QP_B1);                           \
        const float rb2 = CELLS_DPP_F(rc, QP_B2), rb3 = CELLS_DPP_F(rc, QP_B3);                           \
        rr = __builtin_fmaf(rc, rc, rr);                                                                  \
        _Pragma("unroll") for (int i8 = 0; i8 < NS; i8 += 8) {                                            \
            if (i8 < wave_nc) {                                                                           \
                _Pragma("unroll") for (int u = 0; u < 8; ++u) CELLS_CELL(i8 + u, BUF)                     \
            }                                                                                             \
        }                                                                                                 \
    }
                    for (int ch = 0; ch < nchunk; ch += 2) {
                        float rc;
                        wait_all();
                        rc = rnx;
                        lds_barrier();  // chunk ch is in LDS for every wave, everybody is done with chunk ch - 1
                        if (ch + 1 < nchunk) { fetch_ref(ch + 1); stage(1, ch + 1); }
                        CELLS_CHUNK(0)
                        if (ch + 1 < nchunk) {
                            wait_all();
                            rc = rnx;
                            lds_barrier();
                            if (ch + 2 < nchunk) { fetch_ref(ch + 2); stage(0, ch + 2); }
                            CELLS_CHUNK(1)
                        }
                    }
#undef CELLS_CHUNK
#undef CELLS_CELL
                    // |r|^2 of the pixel: the quad's four channel residues
                    rr = rr + CELLS_DPP_F(rr, QP_XOR1);
                    rr = rr + CELLS_DPP_F(rr, QP_XOR2);
                    lds_barrier();  // every wave is done with the ring: the dumps may overwrite it
#pragma unroll
                    for (int i = 0; i < NS; ++i)
                        if (i < wave_nc) *(lds_fw)(size_t)(dump0 + i * 256 + lane * 4) = X[i];
                    lds_wait();

                    // ---- plane loop: planes kw + 4 j + g of the steps of this pass --------------------------
                    const unsigned gA = lds0 + GRAMA_OFF, gB = lds0 + GRAMB_OFF;
#pragma unroll
                    for (int j = 0; j < JS; ++j) {
                        if (j >= jlo && j < jhi) {  // uniform
                            const int info = ki[j];
                            const int tex = info & 0x7ff;
                            const unsigned xa = dump0 + (((unsigned)info >> 11) & 0xfffffu);
                            const bool any = info < 0;
                            float fw = fx[j], fe = 1.0f - fw, fn = fy[j], fs = 1.0f - fn;
                            if (!any) { fw = fw * 0.0f; fe = fe * 0.0f; fn = fn * 0.0f; fs = fs * 0.0f; }  // NaN stays NaN, like ATen
                            const v4f Xc = *(lds_v4f)(size_t)xa;
                            const v4f G00 = *(lds_v4f)(size_t)(gA + tex * 16);
                            const v4f G01 = *(lds_v4f)(size_t)(gA + tex * 16 + 16);
                            const v4f G10 = *(lds_v4f)(size_t)(gA + (tex + pitch) * 16);
                            const float N11 = *(lds_f)(size_t)(gA + (tex + pitch) * 16 + 16);
                            const float D2 = *(lds_f)(size_t)(gB + tex * 4);
                            // |sum_t w_t s_t|^2, separable in the x weights (e, w) and the y weights (s, n)
                            const float ee = fe * fe, ww = fw * fw, ew = fe * fw;
                            const float A = ee * G00.x + ww * G01.x + 2.0f * ew * G00.y;   // top row:    N00, N01, H00
                            const float B = ee * G10.x + ww * N11 + 2.0f * ew * G10.y;     // bottom row: N10, N11, H10
                            const float Cq = ee * G00.z + ww * G01.z + ew * (G00.w + D2);  // cross rows: V00, V01, D1 + D2
                            const float Q = (fs * fs) * A + (fn * fn) * B + 2.0f * (fs * fn) * Cq;
                            const float XW = (fs * fe) * Xc.x + (fs * fw) * Xc.y + (fn * fe) * Xc.z + (fn * fw) * Xc.w;
                            const float c = div_sigma((Q - 2.0f * XW) + rr);
                            float& o = cost[m * JS + j];
                            o = (v == 0) ? (0.0f + c) : (o + c);
                        }
                    }
                    jlo = jhi;
                }  // passes
            }      // windows
        }          // views

        if (bail) {
            if (tid == 0) {
                const int tiles_y = (a.H + TH - 1) / TH;
                tile_flags[b * tiles_x * tiles_y + tile] = 1;
            }
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
                float ssum = 0.0f;
#pragma unroll
                for (int i = 0; i < NWIN * JS; ++i) {
                    const int k = (i / JS) * 4 * JS + (i % JS) * 4 + ge;
                    if (k < a.D) ssum = ssum + expf(cost[i] - mx);
                }
                ssum = ssum + CELLS_DPP_F(ssum, QP_XOR1);
                ssum = ssum + CELLS_DPP_F(ssum, QP_XOR2);
                const float ls = logf(ssum);
                float e = 0.0f;
#pragma unroll
                for (int i = 0; i < NWIN * JS; ++i) {
                    const int k = (i / JS) * 4 * JS + (i % JS) * 4 + ge;
                    if (k < a.D) {
                        const float lp = (cost[i] - mx) - ls;
                        if (logp_out && live) logp_out[obase + (size_t)k * HW] = lp;
                        e = e + dtab[k] * expf(lp);
                    }
                }
                e = e + CELLS_DPP_F(e, QP_XOR1);
                e = e + CELLS_DPP_F(e, QP_XOR2);
                if (depth_out && live && g == 0) depth_out[(size_t)b * HW + p] = e;
            }
        }
        __syncthreads();  // s_item of the next round is visible; nobody still reads this tile's LDS state
        item_par ^= 1;
        item = s_item[item_par];
    }
}

// ---- host side -------------------------------------------------------------------------------------------

namespace {

struct DeviceInfo {
    int n_cu = 0;
    bool lds_raised[2] = {false, false};
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
hipError_t launch_sweep_cells(const SweepArgs& a, void* workspace, hipStream_t stream) {
    const int tiles_x = (a.W + TW - 1) / TW, tiles_y = (a.H + TH - 1) / TH;
    const int tiles = tiles_x * tiles_y;
    const size_t flag_only = ((size_t)a.B * tiles * sizeof(int) + 255) & ~(size_t)255;
    int* flags = reinterpret_cast<int*>(workspace);
    int* queue = reinterpret_cast<int*>(static_cast<char*>(workspace) + flag_only);
    float4* packed = reinterpret_cast<float4*>(static_cast<char*>(workspace) + flag_only + 256);
    hipError_t e = launch_pack_c4(a, workspace, stream);
    if (e != hipSuccess) return e;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    DeviceInfo& di = device_info(dev);
    if (di.n_cu == 0) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        di.n_cu = n;
    }
    const size_t lds = cells_lds_bytes(a.D);
    int nblk = (di.n_cu * 3 + 7) & ~7;  // persistent grid: 3 blocks per CU, a multiple of 8
    const long long full = 8ll * ((tiles + 7) / 8) * a.B;
    if (full <= nblk) nblk = (int)full;
    const int which = a.D <= 64 ? 0 : 1;
    const void* kern = which == 0 ? (const void*)sweep_cells_kernel<1> : (const void*)sweep_cells_kernel<2>;
    if (!di.lds_raised[which]) {
        e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)cells_lds_bytes(128));
        if (e != hipSuccess) return e;
        di.lds_raised[which] = true;
    }
    if (which == 0)
        hipLaunchKernelGGL(sweep_cells_kernel<1>, dim3(nblk), dim3(NT), lds, stream, a, packed, flags, queue, tiles_x, tiles);
    else
        hipLaunchKernelGGL(sweep_cells_kernel<2>, dim3(nblk), dim3(NT), lds, stream, a, packed, flags, queue, tiles_x, tiles);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    return launch_sweep_direct_flagged(a, flags, tiles_x, tiles, stream);
}

}  // namespace pdepth
