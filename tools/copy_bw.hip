// Device-copy bandwidth of this GPU with a plain float4 grid-stride kernel (the
// HBM roofline fraction is quoted against the 8 TB/s spec peak and against what a copy achieves).
// hipcc -O3 --offload-arch=gfx950 -o /tmp/copy_bw tools/copy_bw.hip && /tmp/copy_bw
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256) void copy4(const float4* __restrict__ src, float4* __restrict__ dst, size_t n) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) dst[i] = src[i];
}
__global__ __launch_bounds__(256) void read4(const float4* __restrict__ src, float* __restrict__ out, size_t n) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  float acc = 0.0f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) { const float4 v = src[i]; acc += v.x + v.y + v.z + v.w; }
  if (acc == 12345.678f) out[0] = acc;
}
int main() {
  for (size_t mb : {32, 256, 1024, 4096}) {
    const size_t bytes = mb << 20, n = bytes / sizeof(float4);
    float4 *a, *b; float* o;
    if (hipMalloc(&a, bytes) != hipSuccess || hipMalloc(&b, bytes) != hipSuccess || hipMalloc(&o, 4) != hipSuccess) return 1;
    hipMemset(a, 1, bytes); hipMemset(b, 0, bytes);
    for (int grid : {2048, 8192, 32768}) {
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      const int reps = mb >= 1024 ? 10 : 50;
      hipLaunchKernelGGL(copy4, dim3(grid), dim3(256), 0, 0, a, b, n);
      hipDeviceSynchronize();
      hipEventRecord(e0);
      for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(copy4, dim3(grid), dim3(256), 0, 0, a, b, n);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double copy = 2.0 * bytes * reps / (ms * 1e-3) / 1e9;
      hipEventRecord(e0);
      for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(read4, dim3(grid), dim3(256), 0, 0, a, o, n);
      hipEventRecord(e1); hipEventSynchronize(e1);
      hipEventElapsedTime(&ms, e0, e1);
      printf("%5zu MB  grid %6d  copy (read + write) %7.0f GB/s   read only %7.0f GB/s\n", mb, grid, copy, 1.0 * bytes * reps / (ms * 1e-3) / 1e9);
    }
    hipFree(a); hipFree(b); hipFree(o);
  }
  return 0;
}
