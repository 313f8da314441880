// Wave-level helpers of the persistent sweep kernels: cross-lane scans and reductions, LDS-DMA, packed-fp32 sample
// positions (two planes per instruction, bit-identical to geometry.hpp's scalar chain), the packed cell of a position.
#pragma once
#include <hip/hip_runtime.h>

#include <climits>

#include "geometry.hpp"

namespace pdepth {
namespace wv {

typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));
typedef int v4i __attribute__((ext_vector_type(4)));

constexpr int OOB = 0x7fffffff;        // buffer offset beyond every descriptor: the load returns 0
constexpr int NO_CELL = (int)0x80008000u;   // (cell_x, cell_y of it: -32768, -32768 -- neutral in a max)

__device__ __forceinline__ int opaque_v(int x) { asm volatile("" : "+v"(x)); return x; }
__device__ __forceinline__ int opaque_s(int x) { asm volatile("" : "+s"(x)); return x; }

// workgroup barrier that waits for this wave's LDS traffic only: global loads and stores stay in flight
#define PDEPTH_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

// LDS-DMA (buffer_load ... lds): a wave-instruction moves 16 bytes per active lane from memory to LDS address
// m0 + 16 * lane, no registers in between.  Issued from inline asm (the compiler must not know that these loads write LDS,
// or it drains them in front of the next LDS read); counted in vmcnt like every load: the issuing wave waits for them by hand.
template <typename Rsrc>   // (v4i, or the compiler's own __amdgpu_buffer_rsrc_t: the descriptor the builtin loads of the same view use)
__device__ __forceinline__ void dma_b128(Rsrc rsrc, unsigned lds_addr, int voff, int soff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
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
// min / max over the 64 lanes, the DPP modifier on the min / max itself (the compiler expands update_dpp + min into a move, two
// wait states and the operation per step: 25 instructions per reduction where these take 9); the first step's operand is
// written right in front: a DPP read of a VGPR a vector instruction has just written needs two wait states
#define PDEPTH_WAVE_REDUCE(NAME, OP, SOP)                                                                                  \
    __device__ __forceinline__ int NAME(int v) {                                                                           \
        asm volatile("s_nop 1\n\t" OP " %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"          \
                     OP " %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"                        \
                     OP " %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"                            \
                     OP " %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf" : "+v"(v));                                    \
        return SOP(SOP(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)),                                 \
                   SOP(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));                               \
    }
PDEPTH_WAVE_REDUCE(wave_min_i, "v_min_i32_dpp", min)
PDEPTH_WAVE_REDUCE(wave_max_i, "v_max_i32_dpp", max)
#undef PDEPTH_WAVE_REDUCE

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
// CAUTION (gfx950, ROCm 7.2; measured in sweep_dist.hip, tools/dbg/dist_dbg.py): in a kernel whose other waves run
// v_mfma_f32_16x16x32_f16 on the same SIMD, these packed instructions now and then leave the LOW half of a result
// unwritten in lanes 48..63 (a sample's position then lacks exactly one operation of the chain below: px without K@t,
// gy without the "- cy", ...).  Never in the first pass of a workgroup (every wave of the chip is in its position phase
// then), on any later pass a few hundred samples per launch; a drain of all counters and s_nops in front make no
// difference, the scalar chain of geometry.hpp on the same inputs is always right.  sweep_corr.hip (fp32 matrix
// instructions in the other waves) has run this code through tens of thousands of soak cases without a miss.
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
    // (v_med3_f32: one instruction, no canonicalising v_max in front as fminf / fmaxf get; a NaN comes out as one of the bounds
    //  and is caught by the ordered compare below)
    const int x0 = (int)__builtin_amdgcn_fmed3f(xfl, -2.0f, (float)(W + 1));
    const int y0 = (int)__builtin_amdgcn_fmed3f(yfl, -2.0f, (float)(H + 1));
    const bool any = ix == ix && iy == iy && (unsigned)(x0 + 1) < (unsigned)(W + 1) && (unsigned)(y0 + 1) < (unsigned)(H + 1);
    return any ? (y0 << 16) | (x0 & 0xffff) : NO_CELL;
}
__device__ __forceinline__ int cell_x(int xy) { return (int)(short)(xy & 0xffff); }
__device__ __forceinline__ int cell_y(int xy) { return xy >> 16; }

// x of lane ^ 16 (ds_swizzle: the pattern is part of the instruction) / of the lane whose byte address `a32` holds
// ((lane ^ 32) << 2, computed once): __shfl_xor spends three vector instructions per call on that address
__device__ __forceinline__ float xor16_f(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, x), 0x401f));
}
__device__ __forceinline__ float bperm_f(int a32, float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(a32, __builtin_bit_cast(int, x)));
}
// max without the canonicalising v_max_f32 x, x, x the compiler puts in front of fmaxf (a NaN operand: the other one)
__device__ __forceinline__ float max_raw(float a, float b) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

template <typename T>
__device__ __forceinline__ T kernarg_at(size_t offset) {
    typedef const char __attribute__((address_space(4))) * kptr;
    typedef const volatile T __attribute__((address_space(4))) * vptr;
    return *(vptr)((kptr)__builtin_amdgcn_kernarg_segment_ptr() + offset);
}

}  // namespace wv
}  // namespace pdepth
