// Does a ds_read_b128 with part of the wave masked off cost less LDS time?  (experiment for a tap-reuse sweep)
// build: hipcc -O3 --offload-arch=gfx950 tools/mb_ldsmask.hip -o tools/mb_ldsmask
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4f __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k(float* out, int iters, unsigned long long mask) {
    extern __shared__ float lds[];
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = i;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const bool on = (mask >> lane) & 1ull;
    v4f acc = {0, 0, 0, 0};
    unsigned base = (threadIdx.x * 16u) & 32767u;
    for (int it = 0; it < iters; ++it) {
        if (on) {
            v4f a, b, c, d, e, f, g, h;
            unsigned ad = (base + it * 1024u) & 32767u;
            asm volatile(
                "ds_read_b128 %0, %8\n ds_read_b128 %1, %8 offset:4096\n ds_read_b128 %2, %8 offset:8192\n ds_read_b128 %3, %8 offset:12288\n"
                "ds_read_b128 %4, %8 offset:16384\n ds_read_b128 %5, %8 offset:20480\n ds_read_b128 %6, %8 offset:24576\n ds_read_b128 %7, %8 offset:28672\n"
                "s_waitcnt lgkmcnt(0)\n"
                : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d), "=&v"(e), "=&v"(f), "=&v"(g), "=&v"(h) : "v"(ad));
            acc += a + b + c + d + e + f + g + h;
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
}
int main() {
    float* out; hipMalloc(&out, 1 << 24);
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    struct { const char* n; unsigned long long m; } ms[] = {
        {"all 64 lanes", ~0ull}, {"even lanes", 0x5555555555555555ull}, {"random 50%", 0x9d3a61c7e25b84f6ull},
        {"low 32 lanes", 0xffffffffull}, {"low 16 lanes", 0xffffull}, {"lanes 0-7", 0xffull}, {"every 4th", 0x1111111111111111ull},
        {"random 25%", 0x8102408421004812ull}, {"pairs (2 of 4)", 0x3333333333333333ull}, {"octets alt", 0x00ff00ff00ff00ffull}};
    const int iters = 4096, blocks = 256 * 2;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (auto& m : ms) {
        float best = 1e9;
        for (int r = 0; r < 4; ++r) {
            hipEventRecord(a); hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 65536, 0, out, iters, m.m); hipEventRecord(b); hipEventSynchronize(b);
            float t; hipEventElapsedTime(&t, a, b); if (r && t < best) best = t;
        }
        int n = __builtin_popcountll(m.m);
        printf("%-16s active %2d: %.3f ms  -> %.1f TB/s of active-lane bytes, %.2f clk per wave-instruction per CU\n", m.n, n, best,
               (double)blocks * 4 * iters * 8 * n * 16 / best / 1e9, best * 1e-3 * 2.4e9 / ((double)2 * 4 * iters * 8));
    }
    return 0;
}
