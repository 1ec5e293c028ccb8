// Generalized advantage estimation for the on-policy agents (include/ois.h ois_gae).
//
// Reference: src/my_agents/ppo.py:345-358 (compute_gae, a Python loop of ~8 torch launches
// per rollout step) and src/my_agents/ppoc.py:325-336. Here: one thread per arena walks its
// column of the [T, N] rollout tensors backwards; consecutive lanes read consecutive arenas,
// so every access is a full 256-byte wave row. HBM-bound: 9 B read + 8 B written per entry
// (+ 8 B for the in-place normalisation pass of the ppo mode).
//
// Arithmetic: -ffp-contract=off, every operation rounded on its own, in the reference's
// evaluation order (the oracle oracle/ois_oracle.c:orc_gae spells the same sequence).
//
// At the reference's own rollout lengths (ppo.py:23 rollouts = 16, ppoc.py:23: 20) the work is 27 MB at 65 536 arenas and the
// launches cost more than the bytes: the scan is 4.5 us. So the normalisation is TWO launches, not three: every block of the
// normalise kernel sums the per-wave moments itself (16 KB out of L2, the same fixed order as the separate moments kernel,
// which stays for grids with more than kFoldParts waves). ONE launch was built and measured in round 4 (a lane keeps its T <= 32
// advantages in registers across a grid-wide barrier, then writes them normalised) and is slower: the 8 XCDs' L2s are not
// coherent with each other, a word crosses between them in ~2 us, and any barrier needs several such hops — 102 us with an
// arrival counter (1024 same-address atomics serialise at the memory side), 43 us with per-wave flags under release / acquire
// (an agent-scope release writes back the XCD's whole dirty L2), 24 us with relaxed agent-scope stores and explicit waits,
// against 4 us for the hardware's own barrier, the end of a kernel (profiles/r4_gae_probe.json).
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

#include "../../include/ois.h"

extern "C" int ois_fail_(int code, const char* msg);  // ois_kernels.hip: sets ois_last_error()

namespace {

constexpr int kGaeBlock = 64;   // one wave per block: 65 536 arenas spread over all 1024 SIMDs
constexpr int kNormBlock = 256;
constexpr int kChunk = 8;      // rollout steps per pipeline stage: 24 loads per lane, two stages in flight

// wave64 sum in a fixed order (xor butterfly): every lane ends with the same value
__device__ inline double wave_sum(double x) {
  for (int m = 32; m >= 1; m >>= 1) x += __shfl_xor(x, m, 64);
  return x;
}

template <int MODE>
__global__ __launch_bounds__(kGaeBlock) void gae_scan_kernel(
    int T, int N, const float* __restrict__ rew, const uint8_t* __restrict__ term,
    const float* __restrict__ val, const float* __restrict__ last, float gamma, float lam_or_gl,
    float* __restrict__ ret, float* __restrict__ adv, double* __restrict__ partials) {
  const int i = blockIdx.x * kGaeBlock + threadIdx.x;
  double s1 = 0.0, s2 = 0.0;
  if (i < N) {
    float nv = last[i];
    float a = 0.0f;
    // software pipeline: the loads of chunk c+1 are in flight while chunk c is computed and
    // stored (one wave per SIMD at 65 536 arenas: nothing else hides the HBM latency)
    float r[kChunk], v[kChunk], rn[kChunk], vn[kChunk];
    uint8_t d[kChunk], dn[kChunk];
    auto fetch = [&](int hi, float* fr, float* fv, uint8_t* fd) {
#pragma unroll
      for (int k = 0; k < kChunk; ++k) {
        const int t = hi - 1 - k;
        if (t >= 0) {
          const size_t o = (size_t)t * N + i;
          fr[k] = rew[o]; fv[k] = val[o]; fd[k] = term[o];
        }
      }
    };
    fetch(T, r, v, d);
    for (int hi = T; hi > 0; hi -= kChunk) {
      if (hi > kChunk) fetch(hi - kChunk, rn, vn, dn);
#pragma unroll
      for (int k = 0; k < kChunk; ++k) {
        const int t = hi - 1 - k;
        if (t >= 0) {
          const float nd = d[k] ? 0.0f : 1.0f;
          if (MODE == OIS_GAE_PPO) {
            // rewards[i] - values[i] + discount_factor * not_dones[i] * (next_values + lambda_coefficient * advantage)
            const float g = gamma * nd;
            const float inner = nv + lam_or_gl * a;
            a = (r[k] - v[k]) + g * inner;
          } else {
            // delta = rewards[t] + gamma * (1 - terminated[t]) * next_val - values[t]
            // gae = delta + gamma * lam * (1 - terminated[t]) * gae
            const float delta = (r[k] + (gamma * nd) * nv) - v[k];
            a = delta + (lam_or_gl * nd) * a;
          }
          const size_t o = (size_t)t * N + i;
          adv[o] = a;
          ret[o] = a + v[k];
          nv = v[k];
          s1 += (double)a;
          s2 += (double)a * (double)a;
        }
      }
#pragma unroll
      for (int k = 0; k < kChunk; ++k) { r[k] = rn[k]; v[k] = vn[k]; d[k] = dn[k]; }
    }
  }
  if (partials) {
    s1 = wave_sum(s1); s2 = wave_sum(s2);
    if (threadIdx.x == 0) { partials[2 * blockIdx.x + 0] = s1; partials[2 * blockIdx.x + 1] = s2; }
  }
}

// the moments of the normalisation from the per-wave partials: one fixed order (strided accumulate, then a tree), std unbiased
// like torch.std. Called by all kNormBlock threads of a block; every thread returns the same pair.
__device__ inline void moments_of(size_t count, int nparts, const double* __restrict__ partials, float& mean_f, float& den) {
  __shared__ double sh[2][kNormBlock];
  double s1 = 0.0, s2 = 0.0;
  for (int p = threadIdx.x; p < nparts; p += kNormBlock) { s1 += partials[2 * p]; s2 += partials[2 * p + 1]; }
  sh[0][threadIdx.x] = s1; sh[1][threadIdx.x] = s2;
  __syncthreads();
  for (int w = kNormBlock / 2; w >= 1; w >>= 1) {
    if ((int)threadIdx.x < w) { sh[0][threadIdx.x] += sh[0][threadIdx.x + w]; sh[1][threadIdx.x] += sh[1][threadIdx.x + w]; }
    __syncthreads();
  }
  const double n = (double)count;
  const double mean = sh[0][0] / n;
  double var = (sh[1][0] - n * mean * mean) / (n - 1.0);  // NaN for count == 1, like torch.std
  if (var < 0.0) var = 0.0;
  mean_f = (float)mean;
  den = (float)sqrt(var) + 1e-8f;
}

__global__ __launch_bounds__(kNormBlock) void gae_moments_kernel(size_t count, int nparts, const double* __restrict__ partials,
                                                                float* __restrict__ mean_den) {
  float m, d;
  moments_of(count, nparts, partials, m, d);
  if (threadIdx.x == 0) { mean_den[0] = m; mean_den[1] = d; }
}

// advantages = (advantages - mean) / (std + 1e-8), in place, IEEE division. FOLD: the block sums the partials itself
template <bool FOLD>
__global__ __launch_bounds__(kNormBlock) void gae_normalize_kernel(size_t count, int nparts, const double* __restrict__ partials,
                                                                  const float* __restrict__ mean_den, float* __restrict__ adv) {
  float mean_f, den;
  if (FOLD) moments_of(count, nparts, partials, mean_f, den);
  else { mean_f = mean_den[0]; den = mean_den[1]; }
  const size_t stride = (size_t)gridDim.x * kNormBlock;
  for (size_t o = (size_t)blockIdx.x * kNormBlock + threadIdx.x; o < count; o += stride) adv[o] = (adv[o] - mean_f) / den;
}

constexpr int kFoldParts = 2048;  // up to this many per-wave partials (32 KB) every normalise block sums them itself

int gae_blocks(int N) { return (N + kGaeBlock - 1) / kGaeBlock; }

}  // namespace

extern "C" {

size_t ois_gae_scratch_bytes(int num_envs) {
  if (num_envs <= 0) return 0;
  return ((size_t)gae_blocks(num_envs) * 2 + 1) * sizeof(double);  // per-wave (sum, sum of squares) + (mean, denominator)
}

int ois_gae(int mode, int T, int num_envs, const float* rewards, const uint8_t* terminated, const float* values,
            const float* last_values, double gamma, double lambda, int normalize, float* returns_out,
            float* advantages_out, void* scratch_device, size_t scratch_bytes, void* stream) {
  if (mode != OIS_GAE_PPO && mode != OIS_GAE_PPOC) return ois_fail_(OIS_EINVAL, "ois_gae: unknown mode");
  if (T <= 0 || num_envs <= 0) return ois_fail_(OIS_EINVAL, "ois_gae: T and num_envs must be >= 1");
  if (!rewards || !terminated || !values || !last_values || !returns_out || !advantages_out)
    return ois_fail_(OIS_EINVAL, "ois_gae: null tensor pointer");
  if (normalize && (!scratch_device || scratch_bytes < ois_gae_scratch_bytes(num_envs)))
    return ois_fail_(OIS_ENOMEM, "ois_gae: normalize needs ois_gae_scratch_bytes() of device scratch");
  if (((uintptr_t)scratch_device & 7) != 0) return ois_fail_(OIS_EINVAL, "ois_gae: scratch must be 8-byte aligned");
  const int blocks = gae_blocks(num_envs);
  double* partials = normalize ? (double*)scratch_device : nullptr;
  hipStream_t s = (hipStream_t)stream;
  const float g = (float)gamma;
  if (mode == OIS_GAE_PPO)
    hipLaunchKernelGGL(gae_scan_kernel<OIS_GAE_PPO>, dim3(blocks), dim3(kGaeBlock), 0, s, T, num_envs, rewards, terminated,
                       values, last_values, g, (float)lambda, returns_out, advantages_out, partials);
  else
    hipLaunchKernelGGL(gae_scan_kernel<OIS_GAE_PPOC>, dim3(blocks), dim3(kGaeBlock), 0, s, T, num_envs, rewards, terminated,
                       values, last_values, g, (float)(gamma * lambda), returns_out, advantages_out, partials);
  if (normalize) {
    const size_t count = (size_t)T * (size_t)num_envs;
    size_t nb = (count + kNormBlock - 1) / kNormBlock;
    if (nb > 2048) nb = 2048;  // 8 blocks per CU, grid-stride
    float* mean_den = (float*)(partials + 2 * (size_t)blocks);
    if (blocks <= kFoldParts) {
      hipLaunchKernelGGL(gae_normalize_kernel<true>, dim3((unsigned)nb), dim3(kNormBlock), 0, s, count, blocks, partials, mean_den, advantages_out);
    } else {
      hipLaunchKernelGGL(gae_moments_kernel, dim3(1), dim3(kNormBlock), 0, s, count, blocks, partials, mean_den);
      hipLaunchKernelGGL(gae_normalize_kernel<false>, dim3((unsigned)nb), dim3(kNormBlock), 0, s, count, blocks, partials, mean_den, advantages_out);
    }
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    char msg[256];
    snprintf(msg, sizeof msg, "ois_gae launch failed: %s", hipGetErrorString(e));
    return ois_fail_(OIS_EHIP, msg);
  }
  return OIS_OK;
}

}  // extern "C"
