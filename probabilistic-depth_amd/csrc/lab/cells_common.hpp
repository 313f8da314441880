// Shared device helpers of the cell-list sweep kernels (sweep_cells.hip, sweep_cells_fast.hip): DPP quad permutes,
// wave reductions with a scalar result, LDS-DMA issued from inline asm with hand-counted waits.
#pragma once
#include <hip/hip_runtime.h>

#include "../kernels.hpp"

namespace pdepth {
namespace {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(3))) v4f* lds_v4f;
typedef const __attribute__((address_space(3))) v4i* lds_v4i;
typedef const __attribute__((address_space(3))) float* lds_f;
typedef __attribute__((address_space(3))) float* lds_fw;
typedef __attribute__((address_space(3))) int* lds_iw;
typedef __attribute__((address_space(3))) short* lds_sw;

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
// s_waitcnt vmcnt(n), wave-uniform run-time n in 0..9 (the immediate must be a constant); larger n wait for 9
__device__ __forceinline__ void wait_but(int n) {
    if (n < 4) {
        if (n < 2) { if (n == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); }
        else { if (n == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); }
    } else if (n < 7) {
        if (n == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else if (n == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    } else {
        if (n == 7) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
        else if (n == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
    }
}
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

// Diagnostic build only (-DCELLS_STAMPS, tools/dbg/cells_stamps.py): cycles per phase, summed over waves, in the
// spare bytes behind the queue counters.  No stamp exists in the product build.
#ifdef CELLS_STAMPS
#define CELLS_STAMP(idx)                                                        \
    {                                                                           \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime();           \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                     \
        stamp_acc[idx] += now_ - stamp_t;                                       \
        stamp_t = now_;                                                         \
    }
#else
#define CELLS_STAMP(idx)
#endif

}  // namespace
}  // namespace pdepth
