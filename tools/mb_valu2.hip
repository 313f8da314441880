// VALU issue throughput per SIMD as a function of waves per SIMD (gfx950): independent and dependent v_fma_f32 chains.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int DEP>
__global__ void k(float* out, int iters) {
    float a[8];
    for (int i = 0; i < 8; ++i) a[i] = threadIdx.x * 0.001f + i;
    const float b = 1.0001f, c = 0.5f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < 8; ++i) a[DEP ? 0 : i] = __builtin_fmaf(a[DEP ? 0 : i], b, c);
    }
    float s = 0; for (int i = 0; i < 8; ++i) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
    float* out; hipMalloc(&out, 1 << 24);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    for (int dep = 0; dep < 2; ++dep)
        for (int wps = 1; wps <= 8; wps *= 2) {   // waves per SIMD: blocks of 256 threads (4 waves = 1 per SIMD), wps blocks per CU
            const int blocks = 256 * wps;
            auto kern = dep ? k<1> : k<0>;
            hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, 10); hipDeviceSynchronize();
            hipEventRecord(e0); hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, iters); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double instr_per_simd = (double)wps * iters * 32;
            printf("%s chain, %d waves/SIMD: %.3f ms, %.2f cycles per VALU instruction per SIMD (2.4 GHz)\n", dep ? "dependent" : "independent", wps, ms, ms * 1e-3 * 2.4e9 / instr_per_simd);
        }
    return 0;
}
