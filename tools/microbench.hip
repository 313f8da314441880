// Microbenchmarks that size the sweep kernels: fp32 VALU issue rate (scalar vs packed fma),
// LDS read rate (b32 / b128), and IEEE divide cost.   hipcc --offload-arch=gfx950 -O3 -o mb microbench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float float2v __attribute__((ext_vector_type(2)));

template <int ILP> __global__ void k_fma(float* out, int iters, float a, float b) {
    float x[ILP];
#pragma unroll
    for (int i = 0; i < ILP; ++i) x[i] = threadIdx.x * 0.001f + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < ILP; ++i) x[i] = __builtin_fmaf(x[i], a, b);
    }
    float s = 0; 
#pragma unroll
    for (int i = 0; i < ILP; ++i) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int ILP> __global__ void k_pkfma(float* out, int iters, float a, float b) {
    float2v x[ILP]; float2v av = {a, a * 1.0001f}, bv = {b, b * 0.999f};
#pragma unroll
    for (int i = 0; i < ILP; ++i) x[i] = float2v{threadIdx.x * 0.001f + i, threadIdx.x * 0.002f - i};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < ILP; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(av), "v"(bv));
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < ILP; ++i) s += x[i].x + x[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_div(float* out, int iters, float a, float b) {
    float x[4] = {1.f + threadIdx.x, 2.f + threadIdx.x, 3.f, 4.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) x[i] = (x[i] + a) / (b + x[(i + 1) & 3] * 1e-9f);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x[0] + x[1] + x[2] + x[3];
}
template <int VEC> __global__ void k_lds(float* out, int iters) {
    extern __shared__ float lds[];
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = i;
    __syncthreads();
    float s = 0; int base = threadIdx.x * VEC;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            int idx = (base + j * 1024 + it * VEC * 64) & 8191 & ~(VEC - 1);
            if (VEC == 4) { float4 v = *reinterpret_cast<float4*>(&lds[idx]); s += v.x + v.y + v.z + v.w; }
            else if (VEC == 2) { float2 v = *reinterpret_cast<float2*>(&lds[idx]); s += v.x + v.y; }
            else s += lds[idx];
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <typename F> float timeit(F f, int reps = 5) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < reps; ++r) { hipEventRecord(a); f(); hipEventRecord(b); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b); best = ms < best ? ms : best; }
    return best;
}
int main() {
    float* out; hipMalloc(&out, 1 << 26);
    const int iters = 4096;
    for (int wpb : {64, 256, 512}) for (int bpc : {1, 2, 4, 8}) {
        int blocks = 256 * bpc;
        float t = timeit([&] { hipLaunchKernelGGL(k_fma<8>, dim3(blocks), dim3(wpb), 0, 0, out, iters, 1.0001f, 0.5f); });
        double fl = (double)blocks * wpb * iters * 8;
        float t2 = timeit([&] { hipLaunchKernelGGL(k_pkfma<8>, dim3(blocks), dim3(wpb), 0, 0, out, iters, 1.0001f, 0.5f); });
        printf("threads/block %4d blocks/CU %d : v_fma_f32 %.1f Glane-fma/s (%.1f TFLOP/s)   v_pk_fma_f32 %.1f G-instr-lanes/s (%.1f TFLOP/s)\n", wpb, bpc,
               fl / t / 1e6, 2 * fl / t / 1e9, fl / t2 / 1e6, 4 * fl / t2 / 1e9);
    }
    { int blocks = 256 * 8; float t = timeit([&] { hipLaunchKernelGGL(k_div, dim3(blocks), dim3(256), 0, 0, out, 1024, 1.5f, 2.5f); });
      printf("IEEE fp32 divide: %.1f Gdiv/s  (= %.1f lane-cycles each at 78.6T lane-slots/s)\n", (double)blocks * 256 * 1024 * 4 / t / 1e6, 78.6e12 / ((double)blocks * 256 * 1024 * 4 / t * 1e3)); }
    for (int bpc : {1, 2, 4}) {
        int blocks = 256 * bpc;
        float t1 = timeit([&] { hipLaunchKernelGGL(k_lds<1>, dim3(blocks), dim3(256), 32768, 0, out, 2048); });
        float t2 = timeit([&] { hipLaunchKernelGGL(k_lds<2>, dim3(blocks), dim3(256), 32768, 0, out, 2048); });
        float t4 = timeit([&] { hipLaunchKernelGGL(k_lds<4>, dim3(blocks), dim3(256), 32768, 0, out, 2048); });
        double n = (double)blocks * 256 * 2048 * 8 * 4;
        printf("LDS read, 256 thr x %d blocks/CU: b32 %.1f TB/s  b64 %.1f TB/s  b128 %.1f TB/s\n", bpc, n / t1 / 1e9, 2 * n / t2 / 1e9, 4 * n / t4 / 1e9);
    }
    return 0;
}
