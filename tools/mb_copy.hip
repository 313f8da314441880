// Streaming copy ceiling on the box (what an HBM-bound read+write kernel can reach): 134 MB in, 134 MB out.
// hipcc -O3 --offload-arch=gfx950 tools/mb_copy.hip -o tools/mb_copy
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4f __attribute__((ext_vector_type(4)));
template <int U, bool NT> __global__ __launch_bounds__(256) void k_copy(const v4f* __restrict__ in, v4f* __restrict__ out, size_t n) {
    size_t i = ((size_t)blockIdx.x * U) * 256 + threadIdx.x;
    v4f v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) if (i + u * 256 < n) v[u] = NT ? __builtin_nontemporal_load(in + i + u * 256) : in[i + u * 256];
#pragma unroll
    for (int u = 0; u < U; ++u) if (i + u * 256 < n) { if (NT) __builtin_nontemporal_store(v[u], out + i + u * 256); else out[i + u * 256] = v[u]; }
}
template <int U, bool NT> __global__ __launch_bounds__(256) void k_read(const v4f* __restrict__ in, float* __restrict__ out, size_t n) {
    size_t i = ((size_t)blockIdx.x * U) * 256 + threadIdx.x;
    v4f s = {0, 0, 0, 0};
#pragma unroll
    for (int u = 0; u < U; ++u) if (i + u * 256 < n) s += NT ? __builtin_nontemporal_load(in + i + u * 256) : in[i + u * 256];
    if (s.x + s.y + s.z + s.w == 123.456f) out[0] = 1;
}
int main() {
    const size_t n = (size_t)4 * 64 * 256 * 512 / 4;  // float4 elements of a B=4, D=64, 256x512 volume
    v4f *a, *b; (void)hipMalloc(&a, n * 16); (void)hipMalloc(&b, n * 16); (void)hipMemset(a, 0, n * 16);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    auto run = [&](const char* name, auto launch, double bytes) {
        float best = 1e9;
        for (int r = 0; r < 6; ++r) { (void)hipEventRecord(e0); launch(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); float t; (void)hipEventElapsedTime(&t, e0, e1); if (r && t < best) best = t; }
        printf("%-28s %.1f us  %.2f TB/s\n", name, best * 1e3, bytes / best / 1e9);
    };
    run("copy U=4 plain", [&] { hipLaunchKernelGGL((k_copy<4, false>), dim3((n + 1023) / 1024), dim3(256), 0, 0, a, b, n); }, 2.0 * n * 16);
    run("copy U=4 nontemporal", [&] { hipLaunchKernelGGL((k_copy<4, true>), dim3((n + 1023) / 1024), dim3(256), 0, 0, a, b, n); }, 2.0 * n * 16);
    run("copy U=16 nontemporal", [&] { hipLaunchKernelGGL((k_copy<16, true>), dim3((n + 4095) / 4096), dim3(256), 0, 0, a, b, n); }, 2.0 * n * 16);
    run("read U=16 plain", [&] { hipLaunchKernelGGL((k_read<16, false>), dim3((n + 4095) / 4096), dim3(256), 0, 0, a, (float*)b, n); }, 1.0 * n * 16);
    run("read U=16 nontemporal", [&] { hipLaunchKernelGGL((k_read<16, true>), dim3((n + 4095) / 4096), dim3(256), 0, 0, a, (float*)b, n); }, 1.0 * n * 16);
    run("read U=4 nontemporal", [&] { hipLaunchKernelGGL((k_read<4, true>), dim3((n + 1023) / 1024), dim3(256), 0, 0, a, (float*)b, n); }, 1.0 * n * 16);
    return 0;
}
