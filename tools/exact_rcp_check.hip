// Is the refinement chain of the compiler's own fdiv expansion, without v_div_scale / v_div_fmas /
// v_div_fixup (which only handle exponent extremes), bit-identical to the IEEE 1.0f / x?
// Exhaustive over every float in [2^-40, 2^40) (and the negatives), scalar and packed forms.
// hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -o /tmp/rcpchk tools/exact_rcp_check.hip && /tmp/rcpchk
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
typedef float f2 __attribute__((ext_vector_type(2)));
__device__ inline float lean_rcp(float d) {
  const float r0 = __builtin_amdgcn_rcpf(d);
  const float e = __builtin_fmaf(-d, r0, 1.0f);
  const float r1 = __builtin_fmaf(e, r0, r0);
  const float f2_ = __builtin_fmaf(-d, r1, 1.0f);
  const float q1 = __builtin_fmaf(f2_, r1, r1);
  const float f4 = __builtin_fmaf(-d, q1, 1.0f);
  return __builtin_fmaf(f4, r1, q1);
}
__device__ inline f2 lean_rcp2(f2 d) {
  f2 r0; r0.x = __builtin_amdgcn_rcpf(d.x); r0.y = __builtin_amdgcn_rcpf(d.y);
  const f2 e = __builtin_elementwise_fma(-d, r0, (f2)(1.0f));
  const f2 r1 = __builtin_elementwise_fma(e, r0, r0);
  const f2 g = __builtin_elementwise_fma(-d, r1, (f2)(1.0f));
  const f2 q1 = __builtin_elementwise_fma(g, r1, r1);
  const f2 h = __builtin_elementwise_fma(-d, q1, (f2)(1.0f));
  return __builtin_elementwise_fma(h, r1, q1);
}
__global__ void check(uint32_t lo, uint32_t hi, unsigned long long* bad, uint32_t* first) {
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t b = lo + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; b < hi; b += stride) {
    const uint32_t bits = (uint32_t)b;
    float x = __uint_as_float(bits), nx = -x;
    const float want = 1.0f / x, wantn = 1.0f / nx;
    const float got = lean_rcp(x);
    f2 d; d.x = x; d.y = nx;
    const f2 g2 = lean_rcp2(d);
    if (__float_as_uint(got) != __float_as_uint(want) || __float_as_uint(g2.x) != __float_as_uint(want) ||
        __float_as_uint(g2.y) != __float_as_uint(wantn)) {
      if (atomicAdd(bad, 1ull) == 0) *first = bits;
    }
  }
}
int main() {
  unsigned long long* bad; uint32_t* first;
  hipMalloc(&bad, 8); hipMalloc(&first, 4); hipMemset(bad, 0, 8); hipMemset(first, 0, 4);
  auto bits = [](float f) { uint32_t u; memcpy(&u, &f, 4); return u; };
  const uint32_t lo = bits(0x1p-40f), hi = bits(0x1p40f);
  hipLaunchKernelGGL(check, dim3(4096), dim3(256), 0, 0, lo, hi, bad, first);
  unsigned long long hb; uint32_t hf;
  hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost); hipMemcpy(&hf, first, 4, hipMemcpyDeviceToHost);
  printf("lean reciprocal vs 1.0f/x over %llu floats in [2^-40, 2^40) and their negatives: %llu mismatches", (unsigned long long)(hi - lo), hb);
  if (hb) printf(" (first at bits 0x%08x)", hf);
  printf("\n");
  return hb != 0;
}
