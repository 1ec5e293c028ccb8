// Diagnostic: does a wave64 with only 32 active lanes issue VALU faster on gfx950?
// hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -w -o /tmp/halfwave tools/halfwave_rate.hip && /tmp/halfwave
#include <hip/hip_runtime.h>
#include <cstdio>
template <int HALF>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b) {
  if (HALF && (threadIdx.x & 63) >= 32) return;
  float x[8], y[8], z[8];
  float t = threadIdx.x * 1e-3f;
  for (int j = 0; j < 8; ++j) { x[j] = t + j; y[j] = 1.0f + t * (j + 1) * 1e-3f; z[j] = 0.25f * j + t; }
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int r = 0; r < 8; ++r)
      for (int j = 0; j < 8; ++j) x[j] = __builtin_fmaf(x[j], y[j], z[j]);
  }
  float s = 0;
  for (int j = 0; j < 8; ++j) s += x[j];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int HALF>
void run(const char* name) {
  float* d;
  hipMalloc(&d, 256 * 8 * 256 * sizeof(float));
  const int iters = 4000;
  for (int bpc : {1, 2, 4}) {
    int grid = 256 * bpc;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<HALF>, dim3(grid), dim3(256), 0, 0, d, iters, 1.0001f, 0.5f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<HALF>, dim3(grid), dim3(256), 0, 0, d, iters, 1.0001f, 0.5f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-28s waves/SIMD=%d  %.3f ms  %.3f ns per SIMD wave-instr\n", name, bpc, ms, ms * 1e6 / ((double)iters * 64 * bpc));
  }
  hipFree(d);
}
int main() {
  run<0>("full wave64");
  run<1>("half-active wave (32 lanes)");
  return 0;
}
