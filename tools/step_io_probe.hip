// What the load and store phases of one ois_step launch cost by themselves: 1024 workgroups of one
// wave (one per SIMD, as the step kernel at 65 536 arenas), each lane reads R rows of 16 B and writes W
// rows of 16 B from / to [N][4] float arrays, with a dependent ALU chain of `spin` fmas in between.
// Kernel-only time from events on the dispatch packet.
//   hipcc -O3 --offload-arch=gfx950 -o /tmp/step_io_probe tools/step_io_probe.hip && /tmp/step_io_probe
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <vector>
struct Ptrs { float4* p[16]; };
template <int R, int W>
__global__ __launch_bounds__(64) void io_kernel(Ptrs in, Ptrs out, int n, int spin) {
  const int i = blockIdx.x * 64 + threadIdx.x;
  if (i >= n) return;
  float4 acc = make_float4(1.0f, 2.0f, 3.0f, 4.0f);
#pragma unroll
  for (int r = 0; r < R; ++r) { const float4 t = in.p[r][i]; acc.x += t.x; acc.y += t.y; acc.z += t.z; acc.w += t.w; }
  for (int s = 0; s < spin; ++s) { acc.x = __builtin_fmaf(acc.x, 1.0000001f, acc.y); acc.y = __builtin_fmaf(acc.y, 0.9999999f, acc.x); }
#pragma unroll
  for (int w = 0; w < W; ++w) out.p[w][i] = make_float4(acc.x + w, acc.y, acc.z, acc.w);
}
template <int R, int W>
void run(const char* name, Ptrs a, Ptrs b, int n, int spin, bool cold, float4* trash, size_t trash_n) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  double tot = 0; const int reps = 30;
  for (int r = 0; r < reps + 3; ++r) {
    if (cold) hipMemsetAsync(trash, r, trash_n * sizeof(float4), 0);  // 1 GiB write between launches: cold L2 / MALL
    hipExtLaunchKernelGGL((io_kernel<R, W>), dim3((n + 63) / 64), dim3(64), 0, 0, e0, e1, 0, a, b, n, spin);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (r >= 3) tot += ms;
  }
  printf("%-34s %s spin=%5d : %7.2f us  (%.1f MB)\n", name, cold ? "cold" : "warm", spin, tot / reps * 1e3, (R + W) * 16.0 * n / 1e6);
}
int main() {
  const int n = 65536;
  Ptrs a, b;
  for (int k = 0; k < 16; ++k) { hipMalloc(&a.p[k], (size_t)n * 16); hipMalloc(&b.p[k], (size_t)n * 16); hipMemset(a.p[k], 0, (size_t)n * 16); }
  float4* trash; const size_t trash_n = (1u << 30) / 16; hipMalloc(&trash, trash_n * 16);
  for (int cold = 0; cold < 2; ++cold) {
    run<0, 0>("nothing", a, b, n, 0, cold, trash, trash_n);
    run<16, 0>("16 rows read", a, b, n, 0, cold, trash, trash_n);
    run<0, 14>("14 rows written", a, b, n, 0, cold, trash, trash_n);
    run<16, 14>("16 read + 14 written", a, b, n, 0, cold, trash, trash_n);
    run<13, 12>("13 read + 12 written", a, b, n, 0, cold, trash, trash_n);
    run<16, 14>("16 read + 14 written", a, b, n, 6000, cold, trash, trash_n);
    run<0, 0>("compute only", a, b, n, 6000, cold, trash, trash_n);
  }
  return 0;
}
