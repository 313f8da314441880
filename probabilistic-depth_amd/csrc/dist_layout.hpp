// Layout of the source views for the distance-form sweep kernel (sweep_dist.hip), shared with the kernels that write it
// (pack_dist.hip).
//
// What the sweep evaluates.  est_swp_volume_v4 (warping/homography.py:98-135) computes per pixel and plane
//     cost = | sum_t w_t s_t - r |^2          (img_dis_L2_pard :80-82; t = the four bilinear taps :197, zero outside the image).
// With sum_t w_t = 1 that is, exactly,
//     cost = sum_t w_t |s_t - r|^2  -  sum_{t<t'} w_t w_t' |s_t - s_t'|^2  =  sum_t w_t Y_t - Q.
// Y_t = |s_t - r|^2 for 16 texels x 16 pixels is ONE matrix product (N_t - 2 <s_t, r> + |r|^2 with N_t and |r|^2 riding in
// spare K slots), and Q only depends on the source cell: five squared differences of neighbouring texels, precomputed here.
// Every term is a squared distance: non-negative, of the size of the cost itself once the channel means are removed.
//
// Texels outside the image read zero in the reference (padding_mode='zeros'): the packed image carries a ring of one
// texel holding the zero feature vector (centred: -mu), so the sweep kernel has no border case at all.
//
// Numbers.  A feature x' = (x - mu_c) * 2^e (e per batch item: the sampled maximum lands in [8, 16)) is stored as two
// fp16 values h = fp16(x'), l = fp16(x' - h): h + l = x' to 2^-24 |x'| (fp16 subnormals are honoured by the matrix pipe).
// The kernel multiplies h_s h_r + h_s l_r + l_s h_r with v_mfma_f32_16x16x32_f16 (fp32 accumulation): measured as accurate
// as the fp32 matrix instruction (tools/mb_split16.hip), 4.9 times fewer matrix cycles.
//
// Planes of 16 bytes (8 fp16 = one lane's share of a K = 32 matrix operand) per texel, image (H + 2) x wp(W) (W + 2 rounded up
// to a multiple of 8), stored group-major (below: texel_offset):
//     [hi: 4 NCHK planes]  channels 8 p .. 8 p + 7 of the first 32 NCHK channels, high parts
//     [lo: 4 NCHK planes]  ... low parts
//     [tail: 3 planes]     high parts of the T = C - 32 NCHK left-over channels; their low parts; the specials
//                          (n1, n2, n3, 2^15, 2^4, 2^-7, 0, 0): N = |x'|^2 as three fp16 pieces N = n1 2^15 + n2 2^4 + n3 2^-7,
//                          and the constants that multiply the pixel's pieces of |r'|^2.  As a K = 32 matrix operand the tail is
//                          (high | low | high AGAIN, against the pixel's low parts | specials): the lanes of K slice 2 load
//                          the plane of the high parts a second time (round 5 stored it twice: 16 of 336 bytes per texel)
//     [Q: 1 plane, fp32]   (Dx0, Dy0, Dd, Dx1) of the cell whose top-left texel this is: Dx0 = |s00 - s01|^2,
//                          Dy0 = |s00 - s10|^2, Dd = |s00 - s11|^2 + |s01 - s10|^2, Dx1 = |s10 - s11|^2
//                          (Dy1 = |s01 - s11|^2 is the right neighbour's Dy0)
// NCHK = chunks of 32 channels (0, 1 or 2): C <= 8: 0; C <= 40: 1; C <= 72: 2.
#pragma once
#include <hip/hip_runtime.h>

#include "kernels.hpp"

namespace pdepth {
namespace dist {

constexpr int RING = 1;
constexpr int MAX_C = 72;
constexpr float PIECE_C1 = 32768.0f, PIECE_C2 = 16.0f, PIECE_C3 = 0.0078125f;         // 2^15, 2^4, 2^-7
constexpr float PIECE_I1 = 1.0f / 32768.0f, PIECE_I2 = 1.0f / 16.0f, PIECE_I3 = 128.0f;
constexpr float F16_MAX = 65504.0f;

__host__ __device__ inline int nchk(int C) { return C <= 8 ? 0 : (C <= 40 ? 1 : 2); }
__host__ __device__ inline int nplanes(int C) { return 8 * nchk(C) + 3 + 1; }
// Texel-group-major (round 6): the planes of GROUP = 4 consecutive texels of a row are contiguous -- nplanes x 64 bytes = 1 280
// at C = 67 --, so the 16 texels x 20 planes of an operand block of the sweep (which starts at a group) are 5 KB in ONE piece
// (one DRAM page, one TLB entry) instead of twenty 256-byte pieces 2 MB apart (plane-major, rounds 4-5), and a load instruction
// of the sweep (4 planes of 16 texels) reads four runs of 256 bytes.  Measured against plane-major on one box (profiles/r06_ab/):
// packed sweep -3 % (config 2), -2 % (config 3), -5 % (config 5); with groups of 8 texels (whole 128-byte lines per plane) the
// blocks that start in the middle of a group read half lines: +1 .. +5 % against groups of 4.
constexpr int GROUP = 4;
constexpr int GROUP_PLANE_BYTES = GROUP * 16;
// (rows of whole octets of texels: eight lanes of the pack kernel write two groups -- pack_dist.hip: pack_store_pair)
__host__ __device__ inline int wp(int W) { return (W + 2 * RING + 7) & ~7; }
__host__ __device__ inline int hp(int H) { return H + 2 * RING; }
// bytes of one plane of a view / of one view; a view is followed by 256 bytes that a block of 16 texels starting at the last
// texels of the image may read (never used)
__host__ __device__ inline long long plane_bytes(int H, int W) { return (long long)hp(H) * wp(W) * 16; }
__host__ __device__ inline long long view_bytes(int C, int H, int W) { return (long long)nplanes(C) * plane_bytes(H, W) + 256; }
__host__ __device__ inline int group_bytes(int C) { return nplanes(C) * GROUP_PLANE_BYTES; }
// where the 16 bytes of (plane p, padded texel (yp, xp)) of a view lie
__host__ __device__ inline long long texel_offset(int C, int H, int W, int p, int yp, int xp) {
    return ((long long)yp * (wp(W) / GROUP) + xp / GROUP) * group_bytes(C) + p * GROUP_PLANE_BYTES + (xp % GROUP) * 16;
}

// value = p1 2^15 + p2 2^4 + p3 2^-7 with fp16 pieces (value >= 0, < 2^31)
struct Pieces { _Float16 p1, p2, p3; };
__host__ __device__ inline Pieces split_pieces(float v) {
    Pieces p;
    p.p1 = (_Float16)(v * PIECE_I1);
    const float r1 = v - (float)p.p1 * PIECE_C1;
    p.p2 = (_Float16)(r1 * PIECE_I2);
    const float r2 = r1 - (float)p.p2 * PIECE_C2;
    p.p3 = (_Float16)(r2 * PIECE_I3);
    return p;
}

// Per batch item, from the channel statistics of the pre-pass (sweep_pack.hip: feature_stats_kernel): the power of two
// that scales the centred features, from the largest sampled |x - mu_c| over the channels.  Every kernel that needs it
// evaluates this same function on the same numbers.  amax = max over the channels of (max |x| + |mu_c|) of the sample.
__host__ __device__ inline int scale_exponent(float amax) {
    if (!(amax > 0.0f) || !(amax < 3.0e38f)) return 0;
    int k;
    (void)frexpf(amax, &k);          // amax = f 2^k, f in [0.5, 1)
    int e = 4 - k;                   // amax 2^e in [8, 16)
    return e < -40 ? -40 : (e > 40 ? 40 : e);
}

}  // namespace dist
}  // namespace pdepth
