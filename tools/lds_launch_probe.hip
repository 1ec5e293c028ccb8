// Does allocating LDS cost launch time on gfx950? (diagnostic tool, not product code)
// The same VALU-bound kernel, 1024 single-wave workgroups (one per SIMD), launched back to back
// with 0 B / 768 B / 16 KB of static LDS and with 0 B + a dynamic-LDS launch parameter.
//   hipcc -O3 --offload-arch=gfx950 -w -o /tmp/lds_launch_probe tools/lds_launch_probe.hip && /tmp/lds_launch_probe
#include <hip/hip_runtime.h>
#include <cstdio>

template <int LDS_FLOATS>
__global__ __launch_bounds__(64) void k(float* out, int iters, float a, float b) {
  float x[8];
  for (int j = 0; j < 8; ++j) x[j] = threadIdx.x * 1e-3f + j;
  if constexpr (LDS_FLOATS > 0) {
    __shared__ float s[LDS_FLOATS];
    s[threadIdx.x % LDS_FLOATS] = x[0];
    __syncthreads();
    x[1] += s[(threadIdx.x + 1) % LDS_FLOATS];
  }
  for (int i = 0; i < iters; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) x[j] = __builtin_fmaf(x[j], a, b);
  float t = 0;
  for (int j = 0; j < 8; ++j) t += x[j];
  out[blockIdx.x * 64 + threadIdx.x] = t;
}

__global__ __launch_bounds__(64) void kdyn(float* out, int iters, float a, float b) {
  extern __shared__ float dyn[];
  float x[8];
  for (int j = 0; j < 8; ++j) x[j] = threadIdx.x * 1e-3f + j;
  for (int i = 0; i < iters; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) x[j] = __builtin_fmaf(x[j], a, b);
  float t = 0;
  for (int j = 0; j < 8; ++j) t += x[j];
  if (iters < 0) dyn[threadIdx.x] = t;  // never: keeps the symbol referenced
  out[blockIdx.x * 64 + threadIdx.x] = t;
}

// the LDS-promoted private array of the step kernel also pulled in the dispatch pointer and the
// y / z work-item ids (the promotion pass linearises the 3-D thread id): test those alone
__global__ __launch_bounds__(64) void kdisp(float* out, int iters, float a, float b) {
  const unsigned short* dp = (const unsigned short*)__builtin_amdgcn_dispatch_ptr();
  float x[8];
  for (int j = 0; j < 8; ++j) x[j] = threadIdx.x * 1e-3f + j + (float)dp[2] * 1e-9f;  // workgroup_size_x
  for (int i = 0; i < iters; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) x[j] = __builtin_fmaf(x[j], a, b);
  float t = 0;
  for (int j = 0; j < 8; ++j) t += x[j];
  out[blockIdx.x * 64 + threadIdx.x] = t;
}
__global__ __launch_bounds__(64) void kyz(float* out, int iters, float a, float b) {
  float x[8];
  for (int j = 0; j < 8; ++j) x[j] = (threadIdx.x + threadIdx.y * 3 + threadIdx.z * 5) * 1e-3f + j;
  for (int i = 0; i < iters; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) x[j] = __builtin_fmaf(x[j], a, b);
  float t = 0;
  for (int j = 0; j < 8; ++j) t += x[j];
  out[blockIdx.x * 64 + threadIdx.x] = t;
}
// the real thing: a private array with a run-time index, which AMDGPUPromoteAlloca moves to LDS
__global__ __launch_bounds__(64) void kpromote(float* out, int iters, float a, float b, int sel) {
  float x[8], arr[3];
  for (int j = 0; j < 8; ++j) x[j] = threadIdx.x * 1e-3f + j;
  for (int j = 0; j < 3; ++j) arr[j] = x[j] * 2.0f;
  x[3] += arr[sel % 3] + arr[(sel + 1) % 3];
  for (int i = 0; i < iters; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) x[j] = __builtin_fmaf(x[j], a, b);
  float t = 0;
  for (int j = 0; j < 8; ++j) t += x[j];
  out[blockIdx.x * 64 + threadIdx.x] = t;
}

template <typename F>
void timeit(const char* name, F launch) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 20; ++i) launch();
  hipDeviceSynchronize();
  const int reps = 300;
  hipEventRecord(e0);
  for (int i = 0; i < reps; ++i) launch();
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("%-44s %.2f us per launch\n", name, ms * 1e3 / reps);
}

int main() {
  float* d;
  hipMalloc(&d, 1024 * 64 * sizeof(float));
  for (int iters : {0, 2000}) {  // ~0 and ~35 us of VALU work per launch
    printf("iters = %d\n", iters);
    timeit("no LDS", [&] { hipLaunchKernelGGL(k<0>, dim3(1024), dim3(64), 0, 0, d, iters, 1.0001f, 0.5f); });
    timeit("768 B static LDS (192 floats)", [&] { hipLaunchKernelGGL(k<192>, dim3(1024), dim3(64), 0, 0, d, iters, 1.0001f, 0.5f); });
    timeit("16 KB static LDS", [&] { hipLaunchKernelGGL(k<4096>, dim3(1024), dim3(64), 0, 0, d, iters, 1.0001f, 0.5f); });
    timeit("no static LDS, 768 B dynamic at launch", [&] { hipLaunchKernelGGL(kdyn, dim3(1024), dim3(64), 768, 0, d, iters, 1.0001f, 0.5f); });
    timeit("dispatch pointer read", [&] { hipLaunchKernelGGL(kdisp, dim3(1024), dim3(64), 0, 0, d, iters, 1.0001f, 0.5f); });
    timeit("threadIdx.y / .z read", [&] { hipLaunchKernelGGL(kyz, dim3(1024), dim3(64), 0, 0, d, iters, 1.0001f, 0.5f); });
    timeit("private array, run-time index (promoted)", [&] { hipLaunchKernelGGL(kpromote, dim3(1024), dim3(64), 0, 0, d, iters, 1.0001f, 0.5f, iters & 1); });
    timeit("no LDS (again)", [&] { hipLaunchKernelGGL(k<0>, dim3(1024), dim3(64), 0, 0, d, iters, 1.0001f, 0.5f); });
  }
  return 0;
}
