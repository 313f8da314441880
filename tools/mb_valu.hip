// VALU issue rate on gfx950: scalar v_fma_f32 vs v_pk_fma_f32 (inline asm, nothing for the compiler to pack),
// 8 independent chains per wave, 1..8 waves per SIMD.   hipcc -O3 --offload-arch=gfx950 tools/mb_valu.hip -o tools/mb_valu
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(256) void k_s(float* out, int iters, float a, float b) {
    float x[8];
    for (int i = 0; i < 8; ++i) x[i] = threadIdx.x * 0.001f + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
    }
    float s = 0; for (int i = 0; i < 8; ++i) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void k_p(float* out, int iters, float a, float b) {
    v2f x[8]; v2f av = {a, a * 1.0001f}, bv = {b, b * 0.999f};
    for (int i = 0; i < 8; ++i) x[i] = v2f{threadIdx.x * 0.001f + i, threadIdx.x * 0.002f - i};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(av), "v"(bv));
    }
    float s = 0; for (int i = 0; i < 8; ++i) s += x[i].x + x[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
    float* out; (void)hipMalloc(&out, 1 << 26);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    const int iters = 2048;
    for (int w : {1, 2, 3, 4, 6, 8}) {
        float ts = 1e9, tp = 1e9;
        for (int r = 0; r < 4; ++r) {
            float t;
            (void)hipEventRecord(a); hipLaunchKernelGGL(k_s, dim3(256 * w), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f); (void)hipEventRecord(b); (void)hipEventSynchronize(b);
            (void)hipEventElapsedTime(&t, a, b); if (r && t < ts) ts = t;
            (void)hipEventRecord(a); hipLaunchKernelGGL(k_p, dim3(256 * w), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f); (void)hipEventRecord(b); (void)hipEventSynchronize(b);
            (void)hipEventElapsedTime(&t, a, b); if (r && t < tp) tp = t;
        }
        const double n = 256.0 * w * 256 * iters * 32;  // lane-instructions
        printf("%d waves/SIMD: v_fma_f32 %.1f T lane-instr/s (%.2f clk per wave-instr per SIMD)   v_pk_fma_f32 %.1f T lane-instr/s (%.2f clk)\n", w,
               n / ts / 1e9, ts * 1e-3 * 2.4e9 / (w * iters * 32.0), n / tp / 1e9, tp * 1e-3 * 2.4e9 / (w * iters * 32.0));
    }
    return 0;
}
