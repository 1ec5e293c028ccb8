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
// normalise kernel sums the scan's per-block moments itself (4 KB out of L2 at 65 536 arenas — the scan runs four waves per block up
// to kWideScanArenas arenas; the same fixed order as the separate moments kernel, which stays for grids with more than kFoldParts
// blocks): 9.6-9.7 us at T = 16, 10.7-11.1 at T = 20. ONE launch was built and measured in round 4 (a lane keeps its T <= 32
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

// scan blocks: four waves (one per SIMD of a CU) up to kWideScanArenas arenas — 65 536 arenas are then 256 blocks on the 256 CUs and
// the normalisation folds 256 per-block moments instead of 1024 per-wave ones (T = 16: 11.9 -> 10.1 us for scan + normalise, same
// scan time); one wave per block above, where the wider blocks place worse (524 288 arenas: 23 vs 30 us for the scan)
constexpr int kWideScanArenas = 262144;
constexpr int kNormBlock = 256;
constexpr int kChunk = 8;      // rollout steps per pipeline stage: 24 loads per lane, two stages in flight

// wave64 sum in a fixed order (xor butterfly): every lane ends with the same value
__device__ inline double wave_sum(double x) {
  for (int m = 32; m >= 1; m >>= 1) x += __shfl_xor(x, m, 64);
  return x;
}

template <int MODE, int kGaeBlock>
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
  if (partials) {  // per-block moments: the wave sums, added in a fixed order
    s1 = wave_sum(s1); s2 = wave_sum(s2);
    if constexpr (kGaeBlock == 64) {
      if (threadIdx.x == 0) { partials[2 * blockIdx.x + 0] = s1; partials[2 * blockIdx.x + 1] = s2; }
    } else {
      __shared__ double sh[2][kGaeBlock / 64];
      if ((threadIdx.x & 63) == 0) { sh[0][threadIdx.x >> 6] = s1; sh[1][threadIdx.x >> 6] = s2; }
      __syncthreads();
      if (threadIdx.x == 0) {
        double a = 0.0, b = 0.0;
        for (int w = 0; w < kGaeBlock / 64; ++w) { a += sh[0][w]; b += sh[1][w]; }
        partials[2 * blockIdx.x + 0] = a; partials[2 * blockIdx.x + 1] = b;
      }
    }
  }
}

// (a butterfly inside each wave + one barrier instead of the LDS tree's nine measured 0.3 us SLOWER twice, with 1024 and with 256
// partials: the double-precision shuffles cost more than the barriers)
// the moments of the normalisation from the scan's per-block partials: one fixed order (strided accumulate, then a tree), std unbiased
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
template <bool FOLD, bool VEC>
__global__ __launch_bounds__(kNormBlock) void gae_normalize_kernel(size_t count, int nparts, const double* __restrict__ partials,
                                                                  const float* __restrict__ mean_den, float* __restrict__ adv) {
  float mean_f, den;
  if (FOLD) moments_of(count, nparts, partials, mean_f, den);
  else { mean_f = mean_den[0]; den = mean_den[1]; }
  const size_t stride = (size_t)gridDim.x * kNormBlock;
  // four advantages per lane and access (the host checks the alignment); the division stays the IEEE one, element by element
  const size_t quads = VEC ? count / 4 : 0;
  if (VEC) {
    float4* a4 = reinterpret_cast<float4*>(adv);
    for (size_t o = (size_t)blockIdx.x * kNormBlock + threadIdx.x; o < quads; o += stride) {
      float4 v = a4[o];
      v.x = (v.x - mean_f) / den; v.y = (v.y - mean_f) / den; v.z = (v.z - mean_f) / den; v.w = (v.w - mean_f) / den;
      a4[o] = v;
    }
  }
  for (size_t o = quads * 4 + (size_t)blockIdx.x * kNormBlock + threadIdx.x; o < count; o += stride) adv[o] = (adv[o] - mean_f) / den;
}

constexpr int kFoldParts = 2048;  // up to this many per-block partials (32 KB) every normalise block sums them itself

int gae_block_size(int N) { return N <= kWideScanArenas ? 256 : 64; }
int gae_blocks(int N) { const int bs = gae_block_size(N); return (N + bs - 1) / bs; }

}  // namespace

extern "C" {

size_t ois_gae_scratch_bytes(int num_envs) {
  if (num_envs <= 0) return 0;
  return ((size_t)((num_envs + 63) / 64) * 2 + 1) * sizeof(double);  // per-block (sum, sum of squares), sized for one-wave blocks, + (mean, denominator)
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
  const int bs = gae_block_size(num_envs);
#define OIS_GAE_SCAN(MODE_, BS_, LAM_)                                                                                         \
  hipLaunchKernelGGL((gae_scan_kernel<MODE_, BS_>), dim3(blocks), dim3(BS_), 0, s, T, num_envs, rewards, terminated, values, \
                     last_values, g, (float)(LAM_), returns_out, advantages_out, partials)
  if (mode == OIS_GAE_PPO) { if (bs == 256) OIS_GAE_SCAN(OIS_GAE_PPO, 256, lambda); else OIS_GAE_SCAN(OIS_GAE_PPO, 64, lambda); }
  else { if (bs == 256) OIS_GAE_SCAN(OIS_GAE_PPOC, 256, gamma * lambda); else OIS_GAE_SCAN(OIS_GAE_PPOC, 64, gamma * lambda); }
#undef OIS_GAE_SCAN
  if (normalize) {
    const size_t count = (size_t)T * (size_t)num_envs;
    // four advantages per access from 4 Mi entries on (T = 96 at 65 536 arenas: 32.2 -> 30.7 us; below that it measured equal or slower)
    const bool vec = (((uintptr_t)advantages_out) & 15) == 0 && count >= ((size_t)1 << 22);
    size_t nb = ((vec ? count / 4 : count) + kNormBlock - 1) / kNormBlock;
    if (nb > 1024) nb = 1024;  // 4 blocks per CU, grid-stride (2048: +0.4 us at T = 16, 512: +0.4, 256: +1.6)
    float* mean_den = (float*)(partials + 2 * (size_t)((num_envs + 63) / 64));
#define OIS_GAE_NORM(FOLD_, VEC_) hipLaunchKernelGGL((gae_normalize_kernel<FOLD_, VEC_>), dim3((unsigned)nb), dim3(kNormBlock), 0, s, count, blocks, partials, mean_den, advantages_out)
    if (blocks <= kFoldParts) {
      if (vec) OIS_GAE_NORM(true, true); else OIS_GAE_NORM(true, false);
    } else {
      hipLaunchKernelGGL(gae_moments_kernel, dim3(1), dim3(kNormBlock), 0, s, count, blocks, partials, mean_den);
      if (vec) OIS_GAE_NORM(false, true); else OIS_GAE_NORM(false, false);
    }
#undef OIS_GAE_NORM
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
