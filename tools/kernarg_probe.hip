// What does the SIZE of a kernel's by-value arguments cost per launch? Empty and touch-every-word kernels on a
// 1024 x 64 grid (the step kernel's shape at 65 536 arenas) with 16 B, 784 B (the step kernel's DevConsts + DevBuffers)
// and a pointer to the same 784 B in device memory; begin -> end by events on the dispatch packet (hipExtLaunchKernel).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/kernarg_probe tools/kernarg_probe.hip && /tmp/kernarg_probe
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
struct Big { float f[192]; int n; int pad[3]; };  // 784 B
__global__ void k_small(float* out, int n) { if (n < 0) out[threadIdx.x] = 1.0f; }
__global__ void k_big(Big b, float* out) { if (b.n < 0) out[threadIdx.x] = b.f[threadIdx.x & 127]; }
__global__ void k_big_touch(Big b, float* out) {  // reads every word (uniform: scalar loads), like a kernel that uses its arguments
  float s = 0.0f;
#pragma unroll
  for (int i = 0; i < 192; ++i) s += b.f[i];
  if (s == 12345.0f) out[threadIdx.x] = s;
}
__global__ void k_ptr_touch(const Big* __restrict__ b, float* out) {
  float s = 0.0f;
#pragma unroll
  for (int i = 0; i < 192; ++i) s += b->f[i];
  if (s == 12345.0f) out[threadIdx.x] = s;
}
template <typename F> static double timeit(F launch, int reps) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  double tot = 0.0;
  for (int i = 0; i < reps + 20; ++i) {
    launch(a, b);
    hipEventSynchronize(b);
    float ms = 0.0f;
    hipEventElapsedTime(&ms, a, b);
    if (i >= 20) tot += ms;
  }
  return tot / reps * 1e3;
}
int main() {
  float* out; hipMalloc(&out, 4096);
  Big h{}; for (int i = 0; i < 192; ++i) h.f[i] = 1.0f; h.n = 5;
  Big* d; hipMalloc(&d, sizeof(Big)); hipMemcpy(d, &h, sizeof(Big), hipMemcpyHostToDevice);
  const dim3 g(1024), blk(64);
  printf("empty, 16 B of arguments        : %.2f us\n", timeit([&](hipEvent_t a, hipEvent_t b) { hipExtLaunchKernelGGL(k_small, g, blk, 0, 0, a, b, 0, out, 5); }, 200));
  printf("empty, 784 B by value           : %.2f us\n", timeit([&](hipEvent_t a, hipEvent_t b) { hipExtLaunchKernelGGL(k_big, g, blk, 0, 0, a, b, 0, h, out); }, 200));
  printf("784 B by value, every word read : %.2f us\n", timeit([&](hipEvent_t a, hipEvent_t b) { hipExtLaunchKernelGGL(k_big_touch, g, blk, 0, 0, a, b, 0, h, out); }, 200));
  printf("pointer to 784 B, every word read: %.2f us\n", timeit([&](hipEvent_t a, hipEvent_t b) { hipExtLaunchKernelGGL(k_ptr_touch, g, blk, 0, 0, a, b, 0, (const Big*)d, out); }, 200));
  // the same empty kernel by grid shape: how much of the 3.9 us is per workgroup
  for (int wg : {64, 128, 256, 512, 1024}) {
    const dim3 gg(65536 / wg), bb(wg);
    printf("empty, 65 536 threads as %4d x %4d : %.2f us\n", 65536 / wg, wg,
           timeit([&](hipEvent_t a, hipEvent_t b) { hipExtLaunchKernelGGL(k_small, gg, bb, 0, 0, a, b, 0, out, 5); }, 200));
  }
  for (int nb : {1, 64, 256}) {
    const dim3 gg(nb), bb(64);
    printf("empty, %4d x 64                    : %.2f us\n", nb,
           timeit([&](hipEvent_t a, hipEvent_t b) { hipExtLaunchKernelGGL(k_small, gg, bb, 0, 0, a, b, 0, out, 5); }, 200));
  }
  return 0;
}
