// Micro-kernel for the "epipolar walk" dot phase: acc[J][R] += ref[8ch] . LDS texel(j,r)[8ch]
// 256 threads, per-lane LDS addresses, ds_read_b128 x2 + 8 fma per (j,r).  Measures lane-FMA/s.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int J, int R, int CG>
__global__ __launch_bounds__(256, 2) void k_dots(float* out, int chunks, int WC, int ntex) {
    extern __shared__ __attribute__((aligned(16))) float4 win[];   // [CG][ntex]
    for (int i = threadIdx.x; i < CG * ntex; i += 256) win[i] = make_float4(i * 1e-3f, 1.f, 0.5f, 0.25f);
    __syncthreads();
    const int lx = threadIdx.x & 15, ly = threadIdx.x >> 4;
    float acc[J][R];
#pragma unroll
    for (int j = 0; j < J; ++j)
#pragma unroll
        for (int r = 0; r < R; ++r) acc[j][r] = 0.f;
    for (int ch = 0; ch < chunks; ++ch) {
        float4 rf[CG];
#pragma unroll
        for (int g = 0; g < CG; ++g) rf[g] = make_float4(ch + 1.f, lx * 0.1f, ly * 0.2f, g + 0.3f);
        int base = ly * WC + lx + (ch & 1);
#pragma unroll
        for (int j = 0; j < J; ++j) {
            int rowoff = ((j * 5 + lx) >> 5);           // slowly drifting row (per-lane, data dependent)
            int a = base + rowoff * WC + j;
#pragma unroll
            for (int r = 0; r < R; ++r) {
#pragma unroll
                for (int g = 0; g < CG; ++g) {
                    float4 s = win[g * ntex + a + r * WC];
                    acc[j][r] = __builtin_fmaf(rf[g].x, s.x, acc[j][r]);
                    acc[j][r] = __builtin_fmaf(rf[g].y, s.y, acc[j][r]);
                    acc[j][r] = __builtin_fmaf(rf[g].z, s.z, acc[j][r]);
                    acc[j][r] = __builtin_fmaf(rf[g].w, s.w, acc[j][r]);
                }
            }
        }
        __syncthreads();
    }
    float s = 0;
#pragma unroll
    for (int j = 0; j < J; ++j)
#pragma unroll
        for (int r = 0; r < R; ++r) s += acc[j][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int J, int R, int CG> void run(float* out, const char* nm) {
    int WC = 64, ntex = 64 * 40; size_t lds = (size_t)CG * ntex * 16;
    hipFuncSetAttribute((const void*)k_dots<J, R, CG>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    int chunks = 1024, blocks = 256 * 2;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((k_dots<J, R, CG>), dim3(blocks), dim3(256), lds, 0, out, 8, WC, ntex); hipDeviceSynchronize();
    float best = 1e30f;
    for (int i = 0; i < 3; ++i) { hipEventRecord(a); hipLaunchKernelGGL((k_dots<J, R, CG>), dim3(blocks), dim3(256), lds, 0, out, chunks, WC, ntex); hipEventRecord(b); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b); best = ms < best ? ms : best; }
    double fma = (double)blocks * 256 * chunks * J * R * CG * 4;
    printf("%s J=%d R=%d CG=%d lds=%zuKB: %.2f ms  %.1f T lane-fma/s  (LDS %.1f TB/s) err=%s\n", nm, J, R, CG, lds >> 10, best, fma / best / 1e9, fma * 4 / best / 1e9, hipGetErrorString(hipGetLastError()));
}
int main() {
    float* out; hipMalloc(&out, 1 << 24);
    run<40, 3, 2>(out, "dots"); run<40, 3, 1>(out, "dots"); run<32, 4, 2>(out, "dots"); run<24, 3, 2>(out, "dots"); run<16, 3, 2>(out, "dots");
    return 0;
}
