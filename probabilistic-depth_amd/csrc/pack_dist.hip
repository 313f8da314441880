// Pre-pass of the distance-form sweep kernel (sweep_dist.hip): the source views in the layout of dist_layout.hpp --
// centred, scaled by a power of two, split into fp16 high / low parts in matrix-operand order, with the squared norms
// and the squared neighbour differences the sweep needs per texel -- from the NCHW features (pdepth_sweep_dpv_f32,
// pdepth_pack_source_f32) or straight from the encoder output (pdepth_pack_views_f32: the reference's
// cat(feat, avg_pool2d(rgb)) of models/models.py:518-534 is never materialised).
//
// One lane per column of a strip of four rows of the (H + 2) x (W + 2) image (ring of zero-feature texels = padding_mode
// 'zeros'), channels in groups of eight (= one 16-byte store of high parts and one of low parts per row), the next group's
// 41 loads in flight while a group is converted.  The kernel is a stream: 4 C bytes in, 16 * nplanes bytes out per texel.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>

#include "dist_layout.hpp"
#include "kernels.hpp"
#include "stats_body.hpp"

#ifndef PDEPTH_PACK_NT_LOAD
#define PDEPTH_PACK_NT_LOAD 0
#endif
#ifndef PDEPTH_PACK_FUSE_STATS
#define PDEPTH_PACK_FUSE_STATS 0   // 1: the channel statistics in the first workgroups of the pack kernel (one pre-pass launch instead of two).
                                   // Built, parity-green (graph replays included), measured SLOWER on one box, A/B twice (us per NCHW call,
                                   // separate / fused): B=1 64x128 23.0 / 27.4, B=4 64x128 41.7 / 45.0, B=1 256x512 111.5 / 115.0, headline
                                   // 398 / 400, config 5 4 160 / 4 200 -- every workgroup of the grid polls and reads its means past L2
                                   // (agent-scope loads), which costs more than the launch it saves (profiles/r06_ab/README.md): off
#endif
#ifndef PDEPTH_PACK_NT_STORE
#define PDEPTH_PACK_NT_STORE 0
#endif

namespace pdepth {

namespace {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef int v4i __attribute__((ext_vector_type(4)));

// source accessors: raw feature c of texel (y, x) of view (b, v); the caller only asks for texels inside the image
struct NchwSource {
    const float* base;   // view (b, v)
    int HW, W;
    __device__ __forceinline__ float at(int c, int y, int x) const {
#if PDEPTH_PACK_NT_LOAD
        return __builtin_nontemporal_load(base + (size_t)c * HW + y * W + x);
#else
        return base[(size_t)c * HW + y * W + x];
#endif
    }
};
// the encoder epilogue: channels < Cf from the encoder's feature maps, the rest = avg_pool2d(rgb, rate) as ATen computes it
// (window sum in row-major order, divided by rate^2)
struct ViewSource {
    const float* feat;   // [Cf, H, W] of the view
    const float* im;     // [3, IH, IW]
    int Cf, HW, W, rate, IH, IW;
    __device__ __forceinline__ float at(int c, int y, int x) const {
        if (c < Cf) return feat[(size_t)c * HW + y * W + x];
        const float* p = im + ((size_t)(c - Cf) * IH + (size_t)y * rate) * IW + (size_t)x * rate;
        float sum = 0.0f;
        if (rate == 4 && (IW & 3) == 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float4 q = *reinterpret_cast<const float4*>(p + (size_t)j * IW);
                sum += q.x; sum += q.y; sum += q.z; sum += q.w;
            }
            return sum / 16.0f;
        }
        for (int j = 0; j < rate; ++j)
            for (int i = 0; i < rate; ++i) sum += p[(size_t)j * IW + i];
        return sum / (float)(rate * rate);
    }
};

// mus[c] = mu[c] * 2^e (LDS, zeros beyond C), sc = 2^e.  out = the view's planes.
//
// One wave = 64 consecutive texels of a strip of ROWS rows of the padded image (the strips of a view laid end to end:
// element e = strip * Wp + xp), every lane its column: the lane loads its ROWS + 1 texels of a channel once, the right-hand
// neighbours come from the next lane (DPP wave_shl:1) and, for lane 63, from one extra load per group of eight channels in
// which lane r * 8 + j fetches the neighbour of row r, channel j.  A vector-memory instruction occupies the CU's address
// pipe for ~17 cycles whatever it loads (tools/mb_ta.hip): the first version of this kernel (one thread per texel, its four
// cell corners loaded separately) issued 4 loads per channel and texel, this one (ROWS + 1) / ROWS.  Where a row of the padded image ends inside a wave, the "neighbour" a lane gets belongs to the
// next row's first texel: both lie in the ring (x = W + 1 or beyond, x = -1), both hold the zero feature vector.

// waves per SIMD the pack kernel is compiled for (4: 128 VGPRs)
#ifndef PDEPTH_PACK_OCC
#define PDEPTH_PACK_OCC 4
#endif
#if PDEPTH_PACK_NT_STORE
#define PACK_ST(T, p, v) __builtin_nontemporal_store((v), reinterpret_cast<T*>(p))
#else
#define PACK_ST(T, p, v) (*reinterpret_cast<T*>(p) = (v))
#endif

// Two planes P (even), P + 1 of a lane's texel into the layout of groups of four texels (dist_layout.hpp), in whole 128-byte
// lines.  Lane by lane the two stores would write 64-byte halves of lines (a group's plane is 64 bytes) -- measured 20 % slower
// for the whole kernel than the plane-major layout.  Instead the two quads of every eight lanes trade values (lane ^ 4): the
// first store writes both planes of the EVEN group (lanes 0-3 their own X, lanes 4-7 the Y of lanes 0-3), the second both
// planes of the odd group.  row8 = where plane P of the even group's first texel lies; the eight lanes' texels are in one row.
__device__ __forceinline__ void pack_store_pair(char* row8, int group_bytes, v4i X, v4i Y) {
    const int lane = threadIdx.x & 63;
    const bool odd = lane & 4;
    v4i send, recv;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        send[k] = odd ? X[k] : Y[k];
        recv[k] = __builtin_amdgcn_ds_swizzle(send[k], 0x101f);   // lane ^ 4
    }
    char* p = row8 + (lane & 7) * 16;
    v4i first, second;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        first[k] = odd ? recv[k] : X[k];
        second[k] = odd ? Y[k] : recv[k];
    }
    PACK_ST(v4i, p, first);
    PACK_ST(v4i, p + group_bytes, second);
}

__device__ __forceinline__ float wave_shl1(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, false));
}

// SPLIT (small images: fewer waves than SIMDs -- a wave's chain of 4 NCHK + 1 rounds is then the whole kernel): the four waves of
// a workgroup share one run of 64 texels, wave w converting channel groups w, w + 4, w + 8; the sums over the channels (N and
// the neighbour differences) are added up through LDS (`part`: [4][6][64] floats) in the order of the waves.  Which mode runs
// depends on the image size alone, so the pack of an NCHW call, pdepth_pack_source_f32 and the encoder epilogue agree.
template <int NCHK, int ROWS, bool SPLIT, typename Source>
__device__ __forceinline__ void pack_dist_strip(const Source& src, int C, int H, int W, long long e0, const float* mus, float sc,
                                                char* __restrict__ out, int* __restrict__ item_flags, float* part = nullptr) {
    static_assert(!SPLIT || ROWS == 1, "the split mode is written for one row per strip");
    const int lane = threadIdx.x & 63;
    const int Wp = dist::wp(W), Hp = dist::hp(H);
    const long long nelem = (long long)((Hp + ROWS - 1) / ROWS) * Wp;
    // this lane's column, and the column whose texels lane 63 needs as right-hand neighbours (the "halo": element e0 + 64)
    const long long e = e0 + lane;
    const int strip = (int)(e / Wp), xp = (int)(e - (long long)strip * Wp), yp0 = strip * ROWS;
    const long long eh = e0 + 64;
    const int strip_h = (int)(eh / Wp), xh = (int)(eh - (long long)strip_h * Wp) - dist::RING, yh0 = strip_h * ROWS;
    const int x = xp - dist::RING;
    const bool xin = (unsigned)x < (unsigned)W;
    const int xa = min(max(x, 0), W - 1);
    bool in[ROWS + 1];
    int ya[ROWS + 1];
#pragma unroll
    for (int r = 0; r <= ROWS; ++r) {
        const int y = yp0 + r - dist::RING;
        in[r] = xin && (unsigned)y < (unsigned)H;
        ya[r] = min(max(y, 0), H - 1);
    }
    // halo lane r * 8 + j: row r of the halo column, channel j of the group
    const int hr = min(lane >> 3, ROWS), hj = lane & 7;
    const int yh = yh0 + hr - dist::RING;
    const bool hin = (unsigned)xh < (unsigned)W && (unsigned)yh < (unsigned)H && lane < 8 * (ROWS + 1);
    const int xha = min(max(xh, 0), W - 1), yha = min(max(yh, 0), H - 1);

    constexpr int NG = 4 * NCHK + 1;   // groups of 8 channels; the last one is the tail
    float n[ROWS], dx[ROWS + 1], dy0[ROWS], dd[ROWS];
#pragma unroll
    for (int r = 0; r < ROWS; ++r) n[r] = dy0[r] = dd[r] = 0.f;
#pragma unroll
    for (int r = 0; r <= ROWS; ++r) dx[r] = 0.f;
    bool ovf = false;
    constexpr int NV = 8 * (ROWS + 1) + 1;
    auto issue = [&](int g, float(&v)[NV]) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = min(8 * g + j, C - 1);   // (channels beyond C: loaded from the last one, their mean is zero and so are they)
#pragma unroll
            for (int r = 0; r <= ROWS; ++r) v[r * 8 + j] = src.at(c, ya[r], xa);
        }
        v[NV - 1] = src.at(min(8 * g + hj, C - 1), yha, xha);
    };
    constexpr bool PAIRED = !SPLIT;   // (pack_store_pair)
    auto row8 = [&](int pl, int r) { return out + dist::texel_offset(C, H, W, pl, yp0 + r, xp & ~7); };
    const int GB = dist::group_bytes(C);
    h8 hk[ROWS], lk[ROWS];   // (PAIRED: the planes of the even round, kept for the odd one)
    auto finish = [&](int g, const float(&v)[NV]) {
        h8 hh[ROWS], ll[ROWS];
        // x' = (x - mu) 2^e in one rounding (the scaling is exact); outside the image, and beyond C: x = 0
        const bool hhas = 8 * g + hj < C;
        const float hc = __builtin_fmaf(hhas && hin ? v[NV - 1] : 0.f, sc, -mus[min(8 * g + hj, dist::MAX_C + 7)]);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = 8 * g + j;
            const bool has = c < C;   // uniform
            const float m = mus[min(c, dist::MAX_C + 7)];
            float s[ROWS + 1], sr[ROWS + 1];
#pragma unroll
            for (int r = 0; r <= ROWS; ++r) {
                s[r] = __builtin_fmaf(has && in[r] ? v[r * 8 + j] : 0.f, sc, -m);
                sr[r] = wave_shl1(s[r]);
                const float h63 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, hc), r * 8 + j));
                sr[r] = lane == 63 ? h63 : sr[r];
                const float a = s[r] - sr[r];
                dx[r] = __builtin_fmaf(a, a, dx[r]);
            }
#pragma unroll
            for (int r = 0; r < ROWS; ++r) {
                ovf = ovf || !(fabsf(s[r]) <= 65000.0f);
                const float sx = fminf(fmaxf(s[r], -65000.0f), 65000.0f);
                const _Float16 h = (_Float16)sx;
                hh[r][j] = h;
                ll[r][j] = (_Float16)(sx - (float)h);
                n[r] = __builtin_fmaf(s[r], s[r], n[r]);
                const float b = s[r] - s[r + 1], d1 = s[r] - sr[r + 1], d2 = sr[r] - s[r + 1];
                dy0[r] = __builtin_fmaf(b, b, dy0[r]);
                dd[r] = __builtin_fmaf(d1, d1, dd[r]);
                dd[r] = __builtin_fmaf(d2, d2, dd[r]);
            }
        }
        if (e < nelem) {
#pragma unroll
            for (int r = 0; r < ROWS; ++r) {
                if (yp0 + r >= Hp) break;
                auto at = [&](int pl) { return out + dist::texel_offset(C, H, W, pl, yp0 + r, xp); };
                if (PAIRED) {
                    if (g == 4 * NCHK) {   // the tail: high and low parts are neighbours
                        pack_store_pair(row8(8 * NCHK, r), GB, __builtin_bit_cast(v4i, hh[r]), __builtin_bit_cast(v4i, ll[r]));
                    } else if (g & 1) {
                        pack_store_pair(row8(g - 1, r), GB, __builtin_bit_cast(v4i, hk[r]), __builtin_bit_cast(v4i, hh[r]));
                        pack_store_pair(row8(4 * NCHK + g - 1, r), GB, __builtin_bit_cast(v4i, lk[r]), __builtin_bit_cast(v4i, ll[r]));
                    } else {
                        hk[r] = hh[r]; lk[r] = ll[r];
                    }
                } else if (g < 4 * NCHK) {
                    PACK_ST(h8, at(g), hh[r]);
                    PACK_ST(h8, at(4 * NCHK + g), ll[r]);
                } else {
                    PACK_ST(h8, at(8 * NCHK + 0), hh[r]);
                    PACK_ST(h8, at(8 * NCHK + 1), ll[r]);
                }
            }
        }
    };
    float va[NV], vb[NV];
    if (SPLIT) {
        const int g0 = threadIdx.x >> 6;   // this wave's groups: g0, g0 + 4, g0 + 8 (NG <= 9)
        if (g0 < NG) issue(g0, va);
        if (g0 + 4 < NG) issue(g0 + 4, vb);
        __builtin_amdgcn_sched_barrier(0);
        if (g0 < NG) finish(g0, va);
        if (g0 + 8 < NG) issue(g0 + 8, va);
        __builtin_amdgcn_sched_barrier(0);
        if (g0 + 4 < NG) finish(g0 + 4, vb);
        if (g0 + 8 < NG) finish(g0 + 8, va);
        // the waves' sums, added in the order of the waves by wave 0 (the same bits whatever the timing)
        float* mine = part + (size_t)g0 * 6 * 64 + lane;
        mine[0] = n[0]; mine[64] = dx[0]; mine[128] = dx[1]; mine[192] = dy0[0]; mine[256] = dd[0]; mine[320] = ovf ? 1.0f : 0.0f;
        __syncthreads();
        if (g0 != 0) return;
#pragma unroll
        for (int w = 1; w < 4; ++w) {
            const float* o = part + (size_t)w * 6 * 64 + lane;
            n[0] = n[0] + o[0]; dx[0] = dx[0] + o[64]; dx[1] = dx[1] + o[128]; dy0[0] = dy0[0] + o[192]; dd[0] = dd[0] + o[256];
            ovf = ovf || o[320] != 0.0f;
        }
    } else {
        issue(0, va);
#pragma unroll 1
        for (int g = 0; g < NG; g += 2) {
            if (g + 1 < NG) issue(g + 1, vb);
            __builtin_amdgcn_sched_barrier(0);
            finish(g, va);
            if (g + 1 < NG) {
                if (g + 2 < NG) issue(g + 2, va);
                __builtin_amdgcn_sched_barrier(0);
                finish(g + 1, vb);
            }
        }
    }
    if (e >= nelem) return;
    // specials: N as three fp16 pieces, and the constants that multiply the pixel's pieces of |r'|^2;
    // Q record: (Dx0, Dy0, Dd, Dx1) -- Dx1 of a texel is Dx0 of the texel below
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
        if (yp0 + r >= Hp) break;
        auto at = [&](int pl) { return out + dist::texel_offset(C, H, W, pl, yp0 + r, xp); };
        ovf = ovf || !(n[r] < 2.0e9f);
        const dist::Pieces pn = dist::split_pieces(fminf(n[r], 2.0e9f));
        h8 sp;
        sp[0] = pn.p1; sp[1] = pn.p2; sp[2] = pn.p3;
        sp[3] = (_Float16)dist::PIECE_C1; sp[4] = (_Float16)dist::PIECE_C2; sp[5] = (_Float16)dist::PIECE_C3;
        sp[6] = (_Float16)0.f; sp[7] = (_Float16)0.f;
        if (PAIRED) {
            pack_store_pair(row8(8 * NCHK + 2, r), GB, __builtin_bit_cast(v4i, sp), __builtin_bit_cast(v4i, v4f{dx[r], dy0[r], dd[r], dx[r + 1]}));
        } else {
            PACK_ST(h8, at(8 * NCHK + 2), sp);
            PACK_ST(v4f, at(8 * NCHK + 3), (v4f{dx[r], dy0[r], dd[r], dx[r + 1]}));
        }
    }
    if (ovf) atomicOr(item_flags, 1);
}

// Rows per strip.  Measured on the headline shape (4 views of 256 x 512, C = 67; gpurun_out/pack_time2.txt -> profiles/r05_ab/):
// 1 row: 74 us, 2 rows: 78 us, 4 rows (258 registers, one wave per SIMD): 91 us; the one-thread-per-texel first version
// (4 loads per channel): 70 us.  The kernel moves 140 MB in and 179 MB out: 4.3 TB/s, what a read-and-write stream reaches
// on this memory system -- the address pipe was not the limit after all.  One row keeps the most waves in flight, which is
// what the small shapes want (64 x 128: 17 us against 21 / 31 us).
constexpr int PACK_ROWS = 1;

// the split mode (above): images of up to PDEPTH_PACK_SPLIT_BELOW padded texels
#ifndef PDEPTH_PACK_SPLIT_BELOW
#define PDEPTH_PACK_SPLIT_BELOW 16384
#endif
inline bool pack_dist_split(int H, int W) { return (long long)dist::hp(H) * dist::wp(W) <= PDEPTH_PACK_SPLIT_BELOW; }

__host__ __device__ inline int pack_dist_waves(int H, int W, int ROWS) {
    return (int)(((long long)((dist::hp(H) + ROWS - 1) / ROWS) * dist::wp(W) + 63) / 64);
}

// the scaled channel means of batch item b into LDS (mus: MAX_C + 8 floats), returns 2^e
__device__ __forceinline__ float load_item_scale(const float* __restrict__ stats, int b, float* mus, float* scratch) {
    const float* st = stats + (size_t)b * STATS_STRIDE;
    const int t = threadIdx.x;
    // (agent-scope loads: in the fused pre-pass the rows were written by other workgroups of this launch, on other XCDs)
    auto ld = [&](int i) { return __hip_atomic_load(st + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    float am = t < STATS_VAR ? ld(STATS_AMAX + t) : 0.0f;
    const float mean_t = t < dist::MAX_C + 8 ? ld(t) : 0.0f;   // (asked for with the maxima: one round trip, not two)
    if (t < 64) {
        if (t + 64 < STATS_VAR) am = fmaxf(am, ld(STATS_AMAX + t + 64));
#pragma unroll
        for (int sh = 32; sh >= 1; sh >>= 1) am = fmaxf(am, __shfl_xor(am, sh));
        if (t == 0) scratch[0] = am;
    }
    __syncthreads();
    const float sc = ldexpf(1.0f, dist::scale_exponent(scratch[0]));
    if (t < dist::MAX_C + 8) mus[t] = mean_t * sc;   // (mu = 0 beyond C)
    __syncthreads();
    return sc;
}

__device__ __forceinline__ int xcd_block_order(int nb, int bx) {
    const int xcd = bx & 7, qq = nb >> 3, rr = nb & 7;
    return (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (bx >> 3);
}

// workspace bookkeeping every pack kernel does in its first block: queue counters and slots cleared, layout tag
__device__ __forceinline__ void reset_queue(int* __restrict__ queue) {
    if (threadIdx.x < 64) queue[threadIdx.x] = threadIdx.x == LAYOUT_SLOT ? LAYOUT_DIST16 : 0;
}

// FUSED STATISTICS (tag != 0).  The channel statistics of the call are computed by the first C x B workgroups of this very launch
// (one channel of one batch item each: stats_body.hpp), which publish them with the launch's tag; every workgroup then waits
// for the tags of its batch item before it reads the means and the scale.  Workgroups are dispatched in the order of their
// linear index, so the producers are resident -- and depend on nobody -- before any consumer can occupy their place: the
// launcher fuses only where C x B is a fraction of what the device holds at once.  One lane of a workgroup polls (agent-scope
// acquire loads, s_sleep in between), the others wait at the barrier behind it; a poll that does not come true in ~50 ms
// gives up (the statistics of an earlier call, or zeros, are used: a wrong scale is loud -- NaN items -- and never a hang).
// The tag is a kernel argument: a captured graph replays it.  So the sweep kernel that follows in the same call clears the tags
// (sweep_dist.hip), and only the NCHW sweep entry fuses -- pdepth_pack_source_f32, which no sweep needs to follow, keeps the
// statistics kernel in front.
__device__ __forceinline__ void fused_stats(const float* __restrict__ src, long long bstride, long long vstride, int V, const float* __restrict__ ref,
                                            long long ref_bstride, int B, int C, int H, int W, float* __restrict__ stats, int tag, int b_mine) {
    const int Cs = C < STATS_VAR ? C : STATS_VAR;
    const int lin = blockIdx.y * gridDim.x + blockIdx.x;
    if (lin < Cs * B) {
        const int b = lin / Cs, c = lin - b * Cs;
        float* st = stats + (size_t)b * STATS_STRIDE;
        if (c == 0) {
            if (threadIdx.x < STATS_VAR - C && C + (int)threadIdx.x < STATS_VAR) {   // channels beyond C
                st[C + threadIdx.x] = 0.0f; st[STATS_VAR + C + threadIdx.x] = 0.0f; st[STATS_OFF + C + threadIdx.x] = 0.0f;
                st[STATS_AMAX + C + threadIdx.x] = 0.0f; st[STATS_LAG + C + threadIdx.x] = 0.0f;
            }
            if (threadIdx.x < STATS_NFLAG) reinterpret_cast<int*>(st + STATS_FLAGS)[threadIdx.x] = 0;
        }
        stats_body::channel_stats<false>(src + (size_t)b * bstride + (size_t)c * H * W, vstride, V,
                                         ref ? ref + (size_t)b * ref_bstride + (size_t)c * H * W : nullptr, H, W, 1, W, st + c, st + STATS_VAR + c, 1);
        __syncthreads();   // (thread 0 wrote the channel's row, every thread its share of the zeros above)
        // release without a fence (an agent-scope fence writes back / invalidates a whole L2 -- per workgroup that cost twice what
        // the fusion saves): every thread pushes the words it wrote to the memory side with agent-scope stores, waits for their
        // acknowledgements, and behind a barrier thread 0 publishes the tag -- the guide's "payload, vmcnt(0), flag" hand-off
        if (threadIdx.x == 0) {
            for (int o = 0; o < STATS_FLAGS; o += STATS_VAR) __hip_atomic_store(st + o + c, st[o + c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (c == 0) {
            if (threadIdx.x < STATS_VAR - C && C + (int)threadIdx.x < STATS_VAR)
                for (int o = 0; o < STATS_FLAGS; o += STATS_VAR) __hip_atomic_store(st + o + C + threadIdx.x, 0.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (threadIdx.x < STATS_NFLAG) __hip_atomic_store(reinterpret_cast<int*>(st + STATS_FLAGS) + threadIdx.x, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_store(reinterpret_cast<int*>(st + STATS_READY) + c, tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (threadIdx.x < 64) {
        const int* rdy = reinterpret_cast<const int*>(stats + (size_t)b_mine * STATS_STRIDE + STATS_READY);
        for (int spin = 0; spin < (1 << 16); ++spin) {
            bool ok = true;
            for (int c = threadIdx.x; c < Cs; c += 64) ok = ok && __hip_atomic_load(rdy + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == tag;
            if (__builtin_amdgcn_ballot_w64(!ok) == 0ull) break;
            __builtin_amdgcn_s_sleep(20);
        }
    }
    __syncthreads();   // (the rows behind the tags are read with agent-scope loads: load_item_scale)
}

template <int NCHK, int ROWS, bool SPLIT>
__global__ __launch_bounds__(256, PDEPTH_PACK_OCC) void pack_dist_kernel(const float* __restrict__ src, long long bstride, long long vstride, int V, int C,
                                                        int H, int W, char* __restrict__ out, int* __restrict__ flags, int nflags,
                                                        int* queue, float* __restrict__ stats, const float* __restrict__ ref, long long ref_bstride,
                                                        int B, int tag) {
    for (int i = (blockIdx.y * gridDim.x + blockIdx.x) * 256 + threadIdx.x; i < nflags; i += gridDim.x * gridDim.y * 256) flags[i] = 0;
    if (blockIdx.x == 0 && blockIdx.y == 0) reset_queue(queue);
    const int bv = blockIdx.y, b = bv / V;
    if (tag != 0) fused_stats(src, bstride, vstride, V, ref, ref_bstride, B, C, H, W, stats, tag, b);
    __shared__ float mus[dist::MAX_C + 8];
    __shared__ float scratch[4];
    const float sc = load_item_scale(stats, b, mus, scratch);
    __shared__ float part[SPLIT ? 4 * 6 * 64 : 1];
    const int wave = SPLIT ? (int)blockIdx.x : xcd_block_order(gridDim.x, blockIdx.x) * 4 + (threadIdx.x >> 6);   // (SPLIT: the workgroup's run of texels)
    if (wave >= pack_dist_waves(H, W, ROWS)) return;
    NchwSource s{src + (size_t)b * bstride + (size_t)(bv % V) * vstride, H * W, W};
    pack_dist_strip<NCHK, ROWS, SPLIT>(s, C, H, W, (long long)wave * 64, mus, sc, out + (size_t)bv * dist::view_bytes(C, H, W),
                                       reinterpret_cast<int*>(stats + (size_t)b * STATS_STRIDE + STATS_FLAGS), part);
}

// the encoder epilogue (sweep_pack.hip: pack_views_kernel says what it replaces): views 0..V-1 of an item into the packed
// layout, view V (the reference view) as NCHW [B, Cf + 3, H, W]
template <int NCHK, int ROWS, bool SPLIT>
__global__ __launch_bounds__(256) void pack_views_dist_kernel(const float* __restrict__ feat, const float* __restrict__ rgb, int V, int Cf,
                                                              int H, int W, int rate, int IH, int IW, char* __restrict__ out,
                                                              float* __restrict__ ref_out, int* __restrict__ flags, int nflags, int* queue,
                                                              float* __restrict__ stats) {
    const int HW = H * W, C = Cf + 3;
    for (int i = (blockIdx.y * gridDim.x + blockIdx.x) * 256 + threadIdx.x; i < nflags; i += gridDim.x * gridDim.y * 256) flags[i] = 0;
    if (blockIdx.x == 0 && blockIdx.y == 0) reset_queue(queue);
    const int bv = blockIdx.y, b = bv / (V + 1), v = bv % (V + 1);
    __shared__ float mus[dist::MAX_C + 8];
    __shared__ float scratch[4];
    const float sc = load_item_scale(stats, b, mus, scratch);
    const int pix = xcd_block_order(gridDim.x, blockIdx.x) * 256 + threadIdx.x;
    ViewSource s{feat + (size_t)bv * Cf * HW, rgb + (size_t)bv * 3 * (size_t)IH * IW, Cf, HW, W, rate, IH, IW};
    if (v == V) {   // the reference view: NCHW copy + pooled image
        if (pix >= HW) return;
        const int y = pix / W, x = pix - y * W;
        float* o = ref_out + (size_t)b * C * HW + pix;
        // (sixteen channels' loads in flight, then their stores: a load behind every store is a chain of C round trips --
        //  at 64x128, where a CU holds one such wave, that chain was the kernel: 21 us)
#pragma unroll 1
        for (int c0 = 0; c0 < C; c0 += 16) {
            float t[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) t[j] = s.at(min(c0 + j, C - 1), y, x);
#pragma unroll
            for (int j = 0; j < 16; ++j)
                if (c0 + j < C) o[(size_t)(c0 + j) * HW] = t[j];
        }
        return;
    }
    __shared__ float part[SPLIT ? 4 * 6 * 64 : 1];
    const int wave = SPLIT ? (int)blockIdx.x : xcd_block_order(gridDim.x, blockIdx.x) * 4 + (threadIdx.x >> 6);
    if (wave >= pack_dist_waves(H, W, ROWS)) return;
    pack_dist_strip<NCHK, ROWS, SPLIT>(s, C, H, W, (long long)wave * 64, mus, sc, out + (size_t)(b * V + v) * dist::view_bytes(C, H, W),
                                       reinterpret_cast<int*>(stats + (size_t)b * STATS_STRIDE + STATS_FLAGS), part);
}

// workgroups of pack_dist_kernel<nck, PACK_ROWS, split> the device holds at once (occupancy query, cached)
int pack_dist_resident(int nck, bool split) {
    static int cache[3][2] = {{0, 0}, {0, 0}, {0, 0}};
    int& c = cache[nck][split ? 1 : 0];
    if (c == 0) {
        int per_cu = 0;
        hipError_t e = hipErrorUnknown;
#define PDEPTH_OCC(N, S) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, pack_dist_kernel<N, PACK_ROWS, S>, 256, 0)
        if (nck == 0) { if (split) PDEPTH_OCC(0, true); else PDEPTH_OCC(0, false); }
        else if (nck == 1) { if (split) PDEPTH_OCC(1, true); else PDEPTH_OCC(1, false); }
        else { if (split) PDEPTH_OCC(2, true); else PDEPTH_OCC(2, false); }
#undef PDEPTH_OCC
        c = (e == hipSuccess && per_cu > 0 ? per_cu : 1) * sweep_device_cus();
    }
    return c;
}

}  // namespace

// statistics + packed source of the distance-form kernel (what launch_pack_c4 is for the other two)
hipError_t launch_pack_dist(const SweepArgs& a, void* workspace, hipStream_t stream, bool fuse_stats) {
    int* flags = reinterpret_cast<int*>(workspace);
    char* packed = static_cast<char*>(workspace) + sweep_ws_flag_bytes(a.B, a.H, a.W);
    float* stats = reinterpret_cast<float*>(static_cast<char*>(workspace) + sweep_ws_stats_offset(a.B, a.V, a.C, a.H, a.W));
    int* queue = reinterpret_cast<int*>(static_cast<char*>(workspace) + sweep_ws_flag_only_bytes(a.B, a.H, a.W));
    const int nflags = (int)(sweep_ws_flag_only_bytes(a.B, a.H, a.W) / sizeof(int));
    const bool split = pack_dist_split(a.H, a.W);
    dim3 grid(split ? pack_dist_waves(a.H, a.W, PACK_ROWS) : (pack_dist_waves(a.H, a.W, PACK_ROWS) + 3) / 4, a.B * a.V);
    // one launch where the statistics' producers (C x B workgroups) are a fraction of what the device holds at once and of the
    // grid; else the statistics kernel in front, as through round 5
    const long long nstat = (long long)(a.C < STATS_VAR ? a.C : STATS_VAR) * a.B;
    static std::atomic<unsigned> launches{0};
    int tag = 0;
    // (half of what the occupancy query says the chip holds of this kernel: C x B = 268 at the headline shape, of 1 536)
    if (PDEPTH_PACK_FUSE_STATS && fuse_stats && nstat * 2 <= pack_dist_resident(dist::nchk(a.C), split) && nstat <= (long long)grid.x * grid.y)
        tag = (int)(launches.fetch_add(1) % 0x7fffffffu) + 1;   // (never 0; another one per launch of the process)
    if (tag == 0) {
        hipError_t e = launch_feature_stats(a, stats, stream);
        if (e != hipSuccess) return e;
    }
#define PDEPTH_PACK_DIST(N, R) hipLaunchKernelGGL((pack_dist_kernel<N, R, SPLIT_>), grid, dim3(256), 0, stream, a.src, a.src_bstride, a.src_vstride, a.V, a.C, \
                                                  a.H, a.W, packed, flags, nflags, queue, stats, a.ref, a.ref_bstride, a.B, tag)
#define PDEPTH_PACK_DIST_R(N) do { if (split) { constexpr bool SPLIT_ = true; PDEPTH_PACK_DIST(N, PACK_ROWS); } else { constexpr bool SPLIT_ = false; PDEPTH_PACK_DIST(N, PACK_ROWS); } } while (0)
    switch (dist::nchk(a.C)) {
        case 0: PDEPTH_PACK_DIST_R(0); break;
        case 1: PDEPTH_PACK_DIST_R(1); break;
        default: PDEPTH_PACK_DIST_R(2); break;
    }
#undef PDEPTH_PACK_DIST_R
#undef PDEPTH_PACK_DIST
    return hipGetLastError();
}

hipError_t launch_pack_views_dist(const SweepArgs& a, const float* feat, const float* rgb, int rate, int img_h, int img_w, float* ref_out,
                                  void* workspace, hipStream_t stream) {
    int* flags = reinterpret_cast<int*>(workspace);
    char* packed = static_cast<char*>(workspace) + sweep_ws_flag_bytes(a.B, a.H, a.W);
    float* stats = reinterpret_cast<float*>(static_cast<char*>(workspace) + sweep_ws_stats_offset(a.B, a.V, a.C, a.H, a.W));
    int* queue = reinterpret_cast<int*>(static_cast<char*>(workspace) + sweep_ws_flag_only_bytes(a.B, a.H, a.W));
    hipError_t e = launch_view_stats(a, feat, rgb, rate, img_h, img_w, stats, stream);
    if (e != hipSuccess) return e;
    const int nflags = (int)(sweep_ws_flag_only_bytes(a.B, a.H, a.W) / sizeof(int));
    // (source views: 4 waves of 64 x ROWS texels per block; the reference view: 256 pixels per block)
    const bool split = pack_dist_split(a.H, a.W);
    dim3 grid(std::max(split ? pack_dist_waves(a.H, a.W, PACK_ROWS) : (pack_dist_waves(a.H, a.W, PACK_ROWS) + 3) / 4, (a.H * a.W + 255) / 256),
              a.B * (a.V + 1));
#define PDEPTH_PACK_VIEWS_DIST(N, R) hipLaunchKernelGGL((pack_views_dist_kernel<N, R, SPLIT_>), grid, dim3(256), 0, stream, feat, rgb, a.V, a.C - 3, a.H, a.W, \
                                                        rate, img_h, img_w, packed, ref_out, flags, nflags, queue, stats)
#define PDEPTH_PACK_VIEWS_DIST_R(N) do { if (split) { constexpr bool SPLIT_ = true; PDEPTH_PACK_VIEWS_DIST(N, PACK_ROWS); } else { constexpr bool SPLIT_ = false; PDEPTH_PACK_VIEWS_DIST(N, PACK_ROWS); } } while (0)
    switch (dist::nchk(a.C)) {
        case 0: PDEPTH_PACK_VIEWS_DIST_R(0); break;
        case 1: PDEPTH_PACK_VIEWS_DIST_R(1); break;
        default: PDEPTH_PACK_VIEWS_DIST_R(2); break;
    }
#undef PDEPTH_PACK_VIEWS_DIST_R
#undef PDEPTH_PACK_VIEWS_DIST
    return hipGetLastError();
}

}  // namespace pdepth
