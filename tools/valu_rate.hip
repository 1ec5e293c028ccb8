// VALU issue-rate microbenchmark for gfx950 (diagnostic tool, not product code).
// Measures wall time per wave-instruction per SIMD for several fp32 instruction shapes at
// 1 / 2 / 4 waves per SIMD. Build + run on the GPU box:
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -w -o /tmp/valu_rate tools/valu_rate.hip && /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2v __attribute__((ext_vector_type(2)));

// MODE 0: fma(x, s, s)  one VGPR source      MODE 1: fma(x, y, s) two VGPR sources
// MODE 2: fma(x, y, z)  three VGPR sources   MODE 3: x = fma(y, z, x) (v_fmac shape)
// MODE 4: x = x * y                          MODE 5: dependent chain fma(x, s, s)
// MODE 6: v_pk_fma_f32                       MODE 7: dependent chain fma(x, y, z)
// MODE 8: two interleaved dependent chains   MODE 9: quad_perm DPP add
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b) {
  float t = threadIdx.x * 1e-3f;
  float x[8], y[8], z[8];
  for (int j = 0; j < 8; ++j) { x[j] = t + j; y[j] = 1.0f + t * (j + 1) * 1e-3f; z[j] = 0.25f * j + t; }
  f2v p[8];
  for (int j = 0; j < 8; ++j) p[j] = (f2v){x[j], y[j]};
  f2v A = {a, a}, B = {b, b};
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      if (MODE == 0) { for (int j = 0; j < 8; ++j) x[j] = __builtin_fmaf(x[j], a, b); }
      if (MODE == 1) { for (int j = 0; j < 8; ++j) x[j] = __builtin_fmaf(x[j], y[j], b); }
      if (MODE == 2) { for (int j = 0; j < 8; ++j) x[j] = __builtin_fmaf(x[j], y[j], z[j]); }
      if (MODE == 3) { for (int j = 0; j < 8; ++j) x[j] = __builtin_fmaf(y[j], z[j], x[j]); }
      if (MODE == 4) { for (int j = 0; j < 8; ++j) x[j] = x[j] * y[j]; }
      if (MODE == 5) { for (int j = 0; j < 8; ++j) x[0] = __builtin_fmaf(x[0], a, b); }
      if (MODE == 6) { for (int j = 0; j < 8; ++j) p[j] = __builtin_elementwise_fma(p[j], A, B); }
      if (MODE == 7) { for (int j = 0; j < 8; ++j) x[0] = __builtin_fmaf(x[0], y[j], z[j]); }
      if (MODE == 8) { for (int j = 0; j < 4; ++j) { x[0] = __builtin_fmaf(x[0], y[j], z[j]); x[1] = __builtin_fmaf(x[1], y[j + 4], z[j + 4]); } }
      if (MODE == 9) {
        for (int j = 0; j < 8; ++j)
          x[j] = x[j] + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x[j]), 0xB1, 0xF, 0xF, true));
      }
    }
  }
  float s = 0;
  for (int j = 0; j < 8; ++j) s += x[j] + y[j] + z[j] + p[j].x + p[j].y;
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MODE>
void run(const char* name) {
  float* d;
  hipMalloc(&d, 256 * 4 * 256 * sizeof(float));
  const int iters = 4000;
  for (int bpc : {1, 2, 4}) {
    int grid = 256 * bpc;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, d, iters, 1.0001f, 0.5f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, d, iters, 1.0001f, 0.5f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double ns = ms * 1e6 / ((double)iters * 64 * bpc);
    printf("%-34s waves/SIMD=%d  %.3f ms  %.3f ns per SIMD-instr\n", name, bpc, ms, ns);
  }
  hipFree(d);
}
int main() {
  run<0>("fma(x,s,s) indep");
  run<1>("fma(x,y,s) indep");
  run<2>("fma(x,y,z) indep");
  run<3>("x=fma(y,z,x) (fmac) indep");
  run<4>("x=x*y indep");
  run<6>("v_pk_fma_f32 indep");
  run<5>("fma(x,s,s) dependent chain");
  run<7>("fma(x,y,z) dependent chain");
  run<8>("2 interleaved dependent chains");
  run<9>("x += dpp_quad_perm(x) indep");
  return 0;
}
