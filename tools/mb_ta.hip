// What does a vector-memory instruction cost the texture addresser (TA) of a CU?  16 waves per CU issue one access shape back to
// back on an L2-resident buffer; reported: cycles per wave-instruction per CU at 2.4 GHz nominal (the sweep kernels' shapes).
// Build: hipcc -O3 --offload-arch=gfx950 -o tools/mb_ta tools/mb_ta.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4f __attribute__((ext_vector_type(4)));
typedef int v4i __attribute__((ext_vector_type(4)));

// MODE 0: dwordx4, lane (m = lane & 15, kq = lane >> 4): 4 segments of 256 contiguous bytes, 2 MB apart (the texel operands)
// MODE 1: the same, segment start not a multiple of 64 bytes
// MODE 2: dword, 4 groups of 16 lanes: 64 contiguous bytes each, planes 512 KB apart (reference features, rays)
// MODE 3: dwordx4, 64 lanes contiguous 1 KB
// MODE 4: dword, 64 lanes contiguous 256 B
// MODE 5: dword store, 4 groups of 16 lanes (log-DPV stores)
// MODE 6: dwordx4 store, 64 lanes: 4 groups of 16 lanes x 16 B = 256 B contiguous each
// MODE 7: dwordx4 load, lanes 0..15 only (Q records)
template <int MODE>
__global__ __launch_bounds__(256) void k(float* buf, size_t bytes, int iters, float* out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)buf, 0, (int)bytes, 0x00020000);
    const int m = lane & 15, kq = lane >> 4;
    int voff;
    if (MODE == 0) voff = m * 16 + kq * (2 << 20);
    else if (MODE == 1) voff = m * 16 + kq * (2 << 20) + 16;
    else if (MODE == 2 || MODE == 5) voff = m * 4 + kq * (512 << 10);
    else if (MODE == 3) voff = lane * 16;
    else if (MODE == 4) voff = lane * 4;
    else if (MODE == 6) voff = m * 16 + kq * (512 << 10);
    else voff = lane * 16;
    v4f acc = {0, 0, 0, 0};
    // each wave walks its own region (L2-resident: 8 MB buffer); 8 loads in flight
    int soff = (blockIdx.x * 4 + wave) * 4096 % (1 << 20);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int so = (soff + u * 512) & ((1 << 20) - 1);
            if (MODE == 0 || MODE == 1 || MODE == 3) acc += __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(r, voff, so & ~63, 0));
            else if (MODE == 2 || MODE == 4) acc.x += __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, so & ~63, 0));
            else if (MODE == 5) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, acc.x + it), r, voff, so & ~63, 2);
            else if (MODE == 6) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4i, acc), r, voff, so & ~63, 2);
            else if (lane < 16) acc += __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(r, voff, so & ~63, 0));
        }
        soff += 4096;
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
}
template <int MODE> void run(float* buf, size_t bytes, float* out, const char* nm) {
    const int iters = 2000, blocks = 256 * 4;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, buf, bytes, 10, out); hipDeviceSynchronize();
    float best = 1e30f;
    for (int i = 0; i < 3; ++i) {
        hipEventRecord(a); hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, buf, bytes, iters, out); hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b); best = ms < best ? ms : best;
    }
    const double instr_per_cu = (double)blocks * 4 * iters * 8 / 256.0;
    printf("%-62s %.3f ms  %.1f cycles per wave-instruction per CU (2.4 GHz)\n", nm, best, best * 1e-3 * 2.4e9 / instr_per_cu);
}
int main() {
    float *buf, *out; const size_t bytes = 16 << 20;
    hipMalloc(&buf, bytes); hipMemset(buf, 0, bytes); hipMalloc(&out, 1024 * 256 * 4);
    run<0>(buf, bytes, out, "dwordx4 load, 4 x 256 B segments (texel operands), 64 B aligned");
    run<1>(buf, bytes, out, "dwordx4 load, 4 x 256 B segments, 16 B past a 64 B boundary");
    run<2>(buf, bytes, out, "dword load, 4 x 64 B segments (reference features)");
    run<3>(buf, bytes, out, "dwordx4 load, 1 KB contiguous");
    run<4>(buf, bytes, out, "dword load, 256 B contiguous");
    run<5>(buf, bytes, out, "dword store nt, 4 x 64 B segments (log-DPV)");
    run<6>(buf, bytes, out, "dwordx4 store nt, 4 x 256 B segments");
    run<7>(buf, bytes, out, "dwordx4 load, lanes 0..15 only (Q records)");
    return 0;
}
