// Calibrates rocprofv3 FETCH_SIZE / WRITE_SIZE on gfx950 for the access widths the sweep kernels use
// (MI355X_MICROARCH.md: FETCH_SIZE is exact only after a x2 correction for 16 B/lane streams; other
// widths must be calibrated on a known byte count).  Streams a 1 GiB buffer once per kernel.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void read4(const float* p, float* out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; float s = 0;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) s += p[i];
    if (s == 123.456f) out[0] = s;
}
__global__ void read16(const float4* p, float* out, size_t n4) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; float s = 0;
    for (; i < n4; i += (size_t)gridDim.x * blockDim.x) { float4 v = p[i]; s += v.x + v.y + v.z + v.w; }
    if (s == 123.456f) out[0] = s;
}
__global__ void write4(float* p, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = 1.0f;
}
int main() {
    size_t bytes = 1ull << 30, n = bytes / 4; float *p, *o;
    hipMalloc(&p, bytes); hipMalloc(&o, 64); hipMemset(p, 0, bytes);
    hipLaunchKernelGGL(read4, dim3(2048), dim3(256), 0, 0, p, o, n);
    hipLaunchKernelGGL(read16, dim3(2048), dim3(256), 0, 0, (const float4*)p, o, n / 4);
    hipLaunchKernelGGL(write4, dim3(2048), dim3(256), 0, 0, p, n);
    hipDeviceSynchronize(); printf("streamed %zu bytes per kernel\n", bytes); return 0;
}
