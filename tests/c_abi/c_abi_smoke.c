/* The C ABI of include/ois.h used from plain C, without Python or torch: hipMalloc'ed
 * workspace and actions, create -> reset -> steps -> rollout -> buffers -> stats -> state
 * save / load -> destroy. Built and run by tests/test_gpu_env_api.py::test_c_abi_from_plain_c.
 *   hipcc -x c -o c_abi_smoke c_abi_smoke.c -I../../include -L<lib dir> -lois_hip -Wl,-rpath,<lib dir>
 */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "ois.h"

#define CHECK(x)                                                                              \
  do {                                                                                        \
    int rc_ = (x);                                                                            \
    if (rc_ != 0) { fprintf(stderr, "%s -> %d: %s\n", #x, rc_, ois_last_error()); return 1; } \
  } while (0)
#define HIP(x)                                                                                        \
  do {                                                                                                \
    hipError_t e_ = (x);                                                                              \
    if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; }       \
  } while (0)

int main(int argc, char** argv) {
  const int N = 1000;
  const int device = argc > 1 ? atoi(argv[1]) : 0; /* the ordinal the environment lives on (workspace, stream, actions) */
  int ndev = 0;
  HIP(hipGetDeviceCount(&ndev));
  if (device < 0 || device >= ndev) { fprintf(stderr, "device %d of %d\n", device, ndev); return 1; }
  HIP(hipSetDevice(device));
  ois_config cfg;
  CHECK(ois_config_default(&cfg));
  cfg.num_envs = N;
  cfg.seed = 5;
  if (ois_abi_version() != OIS_ABI_VERSION) { fprintf(stderr, "ABI mismatch\n"); return 1; }
  size_t bytes = ois_workspace_bytes(&cfg);
  if (bytes == 0) { fprintf(stderr, "bad config: %s\n", ois_last_error()); return 1; }
  void* ws = NULL;
  HIP(hipMalloc(&ws, bytes));
  hipStream_t stream;
  HIP(hipStreamCreate(&stream));
  ois_env* env = NULL;
  if (ois_create(&cfg, ndev, ws, bytes, &env) != OIS_EINVAL) { fprintf(stderr, "device ordinal out of range not refused\n"); return 1; }
  CHECK(ois_create(&cfg, device, ws, bytes, &env));
  if (ois_step(env, (const float*)ws, stream) != OIS_ESTATE) { fprintf(stderr, "step before reset not refused\n"); return 1; }
  CHECK(ois_reset(env, 5, stream));
  char info[256];
  CHECK(ois_kernel_info(env, info, sizeof info)); /* 1000 arenas on the default plane: a lane pair per arena, flat shell */
  if (strncmp(info, "step_kernel<2,1,4,1>", 20) != 0 || !strstr(info, "plane=flat")) { fprintf(stderr, "kernel info: %s\n", info); return 1; }
  float slopes[2] = {0.0f, 0.2f};
  CHECK(ois_randomize_params(env, 0.5f, 1.0f, slopes, 2, stream));
  CHECK(ois_kernel_info(env, info, sizeof info));
  if (!strstr(info, "plane=per-arena")) { fprintf(stderr, "slopes did not leave the flat shell: %s\n", info); return 1; }
  CHECK(ois_randomize_params(env, 0.85f, 0.85f, NULL, 0, stream)); /* friction only: flat again */
  CHECK(ois_kernel_info(env, info, sizeof info));
  if (!strstr(info, "plane=flat")) { fprintf(stderr, "friction-only randomisation left the flat shell: %s\n", info); return 1; }
  if (ois_kernel_info(env, NULL, 0) != OIS_EINVAL) { fprintf(stderr, "null kernel-info buffer not refused\n"); return 1; }

  float* h_act = (float*)malloc(sizeof(float) * N * 4);
  for (int i = 0; i < N * 4; ++i) h_act[i] = (float)((i * 37) % 100) / 100.0f;
  float* d_act = NULL;
  HIP(hipMalloc((void**)&d_act, sizeof(float) * N * 4));
  HIP(hipMemcpyAsync(d_act, h_act, sizeof(float) * N * 4, hipMemcpyHostToDevice, stream));
  for (int t = 0; t < 5; ++t) CHECK(ois_step(env, d_act, stream));
  float* d_sum = NULL;
  HIP(hipMalloc((void**)&d_sum, sizeof(float) * N));
  CHECK(ois_rollout(env, 100, OIS_POLICY_RANDOM, NULL, d_sum, stream)); /* crosses the timeout at step 99 */

  void* p = NULL;
  size_t n = 0;
  CHECK(ois_buffer(env, OIS_BUF_OBS, &p, &n));
  if (n != sizeof(float) * N * 6 || (char*)p < (char*)ws || (char*)p + n > (char*)ws + bytes) { fprintf(stderr, "obs buffer not inside the workspace\n"); return 1; }
  float* h_obs = (float*)malloc(n);
  HIP(hipMemcpyAsync(h_obs, p, n, hipMemcpyDeviceToHost, stream));
  CHECK(ois_buffer(env, OIS_BUF_COUNTERS, &p, &n));
  int* h_cnt = (int*)malloc(n);
  HIP(hipMemcpyAsync(h_cnt, p, n, hipMemcpyDeviceToHost, stream));
  float* h_sum = (float*)malloc(sizeof(float) * N);
  HIP(hipMemcpyAsync(h_sum, d_sum, sizeof(float) * N, hipMemcpyDeviceToHost, stream));
  ois_stats st;
  CHECK(ois_stats_fetch(env, &st, 0, stream)); /* synchronises the stream */
  for (int i = 0; i < N * 6; ++i)
    if (!isfinite(h_obs[i])) { fprintf(stderr, "non-finite observation\n"); return 1; }
  for (int i = 0; i < N; ++i) {
    if (h_cnt[4 * i + 1] != 105 || h_cnt[4 * i + 2] != 2 || h_cnt[4 * i + 0] != 105 - 99) { fprintf(stderr, "counters of arena %d: %d %d %d\n", i, h_cnt[4 * i], h_cnt[4 * i + 1], h_cnt[4 * i + 2]); return 1; }
    if (!isfinite(h_sum[i])) { fprintf(stderr, "non-finite reward sum\n"); return 1; }
  }
  if (st.episodes != (double)N || st.steps != (double)N * 105.0) { fprintf(stderr, "stats: %f episodes, %f steps\n", st.episodes, st.steps); return 1; }

  /* the step's logs through pinned host memory (ois_logs_*): unpublished state -> the fetch enqueues the finalisation
   * itself; with ois_logs_enable the step does, and the fetch only waits for the sequence number */
  float ei[OIS_EPISODE_INFO_LEN], diag[OIS_ENV0_DIAG_LEN], ei_dev[OIS_EPISODE_INFO_LEN], diag_dev[OIS_ENV0_DIAG_LEN];
  CHECK(ois_logs_fetch(env, ei, diag, stream));
  if (ei[OIS_EI_COUNT] != 0.0f + N || ei[OIS_EI_STEP] != 99.0f) { fprintf(stderr, "logs: count %f at step %f\n", ei[OIS_EI_COUNT], ei[OIS_EI_STEP]); return 1; }
  CHECK(ois_logs_enable(env, 1));
  for (int t = 0; t < 3; ++t) {
    CHECK(ois_step(env, d_act, stream));
    CHECK(ois_logs_fetch(env, ei, diag, stream));
    CHECK(ois_env0_diag(env, diag_dev, stream));
    CHECK(ois_buffer(env, OIS_BUF_EPISODE_INFO, &p, &n));
    HIP(hipMemcpy(ei_dev, p, sizeof ei_dev, hipMemcpyDeviceToHost));
    if (memcmp(ei, ei_dev, sizeof ei) != 0 || memcmp(diag, diag_dev, sizeof diag) != 0) { fprintf(stderr, "published logs differ from the device buffers at step %d\n", t); return 1; }
    if (diag[8] != 2.0f) { fprintf(stderr, "env-0 episode count %f\n", diag[8]); return 1; }
  }
  CHECK(ois_logs_enable(env, 0));
  /* the wrapper's per-step copies in one launch (ois_outputs_copy): fresh arrays of the caller's == the library's buffers */
  {
    float *o_obs = NULL, *o_rew = NULL;
    unsigned char *o_term = NULL, *o_trunc = NULL;
    HIP(hipMalloc((void**)&o_obs, sizeof(float) * N * 6)); HIP(hipMalloc((void**)&o_rew, sizeof(float) * N));
    HIP(hipMalloc((void**)&o_term, N)); HIP(hipMalloc((void**)&o_trunc, N));
    HIP(hipMemsetAsync(o_trunc, 0xff, N, stream));
    CHECK(ois_outputs_copy(env, o_obs, o_rew, o_term, o_trunc, stream));
    float* a = (float*)malloc(sizeof(float) * N * 6); float* b = (float*)malloc(sizeof(float) * N * 6);
    unsigned char* t = (unsigned char*)malloc(N);
    HIP(hipMemcpyAsync(a, o_obs, sizeof(float) * N * 6, hipMemcpyDeviceToHost, stream));
    CHECK(ois_buffer(env, OIS_BUF_OBS, &p, &n));
    HIP(hipMemcpyAsync(b, p, n, hipMemcpyDeviceToHost, stream));
    HIP(hipMemcpyAsync(t, o_trunc, N, hipMemcpyDeviceToHost, stream));
    HIP(hipStreamSynchronize(stream));
    if (memcmp(a, b, sizeof(float) * N * 6) != 0) { fprintf(stderr, "ois_outputs_copy: observations differ\n"); return 1; }
    for (int i = 0; i < N; ++i) if (t[i] != 0) { fprintf(stderr, "ois_outputs_copy: truncated not zeroed\n"); return 1; }
    if (ois_outputs_copy(env, (float*)((char*)o_obs + 4), NULL, NULL, NULL, stream) != OIS_EINVAL) { fprintf(stderr, "misaligned obs_out not refused\n"); return 1; }
    HIP(hipFree(o_obs)); HIP(hipFree(o_rew)); HIP(hipFree(o_term)); HIP(hipFree(o_trunc)); free(a); free(b); free(t);
  }
  /* a caller that writes plane normals itself leaves the flat-plane kernels by saying so */
  CHECK(ois_params_dirty(env));
  CHECK(ois_kernel_info(env, info, sizeof info));
  if (!strstr(info, "plane=per-arena")) { fprintf(stderr, "ois_params_dirty did not leave the flat shell: %s\n", info); return 1; }
  CHECK(ois_randomize_params(env, 0.85f, 0.85f, NULL, 0, stream)); /* back on the default plane */
  if (ois_logs_fetch(env, NULL, diag, stream) != OIS_EINVAL) { fprintf(stderr, "null logs output not refused\n"); return 1; }

  size_t sb = ois_state_bytes(env);
  void* blob = malloc(sb);
  CHECK(ois_save_state(env, blob, sb, stream));
  CHECK(ois_rollout(env, 3, OIS_POLICY_ZERO, NULL, NULL, stream));
  CHECK(ois_load_state(env, blob, sb, stream));
  CHECK(ois_buffer(env, OIS_BUF_COUNTERS, &p, &n));
  HIP(hipMemcpy(h_cnt, p, n, hipMemcpyDeviceToHost));
  if (h_cnt[1] != 108) { fprintf(stderr, "state not restored\n"); return 1; }

  if (ois_rollout(env, 0, OIS_POLICY_ZERO, NULL, NULL, stream) != OIS_EINVAL) { fprintf(stderr, "k = 0 not refused\n"); return 1; }
  ois_destroy(env);
  HIP(hipFree(d_act)); HIP(hipFree(d_sum)); HIP(hipFree(ws));
  free(h_act); free(h_obs); free(h_cnt); free(h_sum); free(blob);
  printf("c abi ok: %d arenas, %.0f episodes, mean return %.3f on device %d\n", N, st.episodes, st.return_sum / st.episodes, device);
  return 0;
}
