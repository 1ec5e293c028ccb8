/*
 * ois_oracle.h — CPU ORACLE (test infrastructure, NOT product code).
 *
 * A scalar, one-arena-at-a-time float32 restatement of the hot path of
 * meesjansen/Options_In_Simulation (src/my_envs/KAMMA.py and variants) plus the
 * build-defined rigid-body model of docs/SPEC.md §3. Only tests/, bench.py's
 * cpu_baseline leg and __graft_entry__.smoke() may load this library; the product
 * (options_in_simulation_amd/) never does.
 *
 * PARITY STATUS
 *   - pinned by the reference's OWN statements executed here, step by step (tests/golden/refexec_cosim.npz and
 *     refexec_blocks.npz, made by tests/golden/make_golden_cosim.py; tests/test_refexec_cosim.py): `TorqueDistributionTask`
 *     of each of the 12 env files is compiled out of the reference's syntax tree WITHOUT the statements that need Isaac
 *     Sim (tests/golden/refexec_task.py: rule R1-R4, the removed lines are listed in refexec_cosim_lines.json; nothing is
 *     written in their place) and run in lock-step with orc_reset / orc_step for 100-200 steps with falls, bound
 *     violations and timeouts: pre_physics_step (controller, gamma, switch / blend, 2-D / 1-D expansion, clip, guiding
 *     reward, the eval sums), is_done, calculate_metrics, reset_idx (counters, small-command mask, ratio sums, extras,
 *     fraction, Rand / Box / Grid / Eval samplers), get_observations with the 26 env-0 diagnostics, the projection tail,
 *     the order of all of it inside post_physics_step. The simulator's outcome, the uniform draws and the three
 *     quat_apply results per step are DATA in the fixture (orc_test_hooks below).
 *   - pinned by whole functions of the reference executed on seeded inputs (tests/golden/refexec_*.npz, made by
 *     tests/golden/make_golden_ref.py; the modules cannot be imported whole — omni / pxr / gym / skrl at module level,
 *     plain ModuleNotFoundError, SURVEY 8c — and no stand-in for an absent library is supplied): TASK_CFG +
 *     update_config of the 12 env files, calculate_metrics (KAMMA.py:767, KA_DDPG_4D, KAMMA_eval), get_observations,
 *     sample_velocity_command's curriculum / EvalSampling / GridSampling branches, PPO's compute_gae, Agent.record_transition,
 *     the terrain generator families and the Terrain class (tests/test_refexec.py).
 *   - left on the build's own transcription (tests/golden/refmath_*.npz, tests/golden/make_golden.py), because the
 *     statement itself is a call into omni.isaac.core.utils.torch, which is not under /root/reference: quat_apply (has_fallen
 *     :749, forward / up :701, :707), quat_mul (the yaw quaternion :434), torch_rand_float (lower + (upper - lower) u,
 *     :424-456, :571), normalize + quat_apply inside quat_apply_yaw (:895-899; the index arithmetic of get_heights,
 *     :882-893, is executed on a field of the reference's own Terrain class: refexec_blocks.npz heights__*).
 *   - physics (PhysX 5 inside Isaac Sim 4.0.0, not in /root/reference, no golden vectors
 *     anywhere in the reference): PARITY UNPINNED. The model is defined by docs/SPEC.md §3;
 *     parity is oracle <-> HIP kernel only.
 *   - RNG: Philox4x32-10 (Salmon et al., SC'11; Random123 v1.14 known-answer vectors are
 *     checked in tests/test_oracle_rng_physics.py). The reference uses torch's global generator,
 *     which cannot be reproduced counter-for-counter; draw ORDER follows KAMMA.py.
 *   - either side of the step, also reference-followed and pinned the same way (torch
 *     transcriptions, bit-exact): orc_gae (src/my_agents/ppo.py:345-358, ppoc.py:325-336 ->
 *     refmath_gae.npz), orc_get_heights (KAMMA.py:872-899 -> refmath_heights.npz); the
 *     transition-memory rows follow train_kaddpg_4d_nofifo_random_RLIL.py:105-206 and
 *     base_KA_DDPG.py:205-224 (no reference vectors: semantics checked by tests/test_replay_memory.py).
 *   - the wheel contact on the height field and the terrain curriculum (docs/SPEC.md §3.8) are
 *     build-defined like the rest of the physics: PARITY UNPINNED.
 *
 * The config struct is the product ABI's (include/ois.h) so that one dict drives both.
 */
#ifndef OIS_ORACLE_H_
#define OIS_ORACLE_H_

#include <stddef.h>
#include <stdint.h>

#include "../include/ois.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_env orc_env;

/* ---- environment-level API (mirrors ois_* but on host memory) ---- */
int orc_create(const ois_config* cfg, orc_env** out);
void orc_destroy(orc_env* env);
int orc_reset(orc_env* env, uint64_t seed);
/* actions: float [N, action_dim] host */
int orc_step(orc_env* env, const float* actions);
/* policy: OIS_POLICY_*; actions [k,N,A] for EXTERNAL; reward_sum [N] optional */
int orc_rollout(orc_env* env, int k, int policy, const float* actions, float* reward_sum);
int orc_randomize_params(orc_env* env, float mu_lo, float mu_hi, const float* slopes, int n_slopes);
/* number of OpenMP threads used by orc_step/orc_rollout (1 = scalar). */
int orc_set_threads(orc_env* env, int threads);

/* Fixture replay (tests/golden/refexec_cosim_*.npz: the reference's own statements run in lock-step with this env).
 * The reference draws from torch's generator and steps PhysX; to compare step by step the env takes those three
 * things from the caller: the uniform of the stochastic switch (KAMMA.py:633), the draws of reset_idx (order of
 * orc_reset_draws; slot 8 = the standard normal of GridSampling), the simulator's outcome, and — for the per-level command
 * curriculum — the normal draw and the terrain level of every arena. Arrays stay the
 * caller's and are read at every orc_reset / orc_step until cleared (NULL). physics rows: COM position 3,
 * quaternion 4 (w,x,y,z), linear velocity 3, angular velocity 3 (world), joint velocities 4, base-link position 3
 * (written only). */
#define ORC_RESET_U_STRIDE 12
#define ORC_PHYSICS_STRIDE 20
typedef struct {
  const float* switch_u;   /* [N] */
  const float* reset_u;    /* [N, ORC_RESET_U_STRIDE] */
  const float* physics_in; /* [N, ORC_PHYSICS_STRIDE]: replaces the outcome of the substeps */
  float* physics_out;      /* [N, ORC_PHYSICS_STRIDE]: receives it (before is_done / reset) */
  const float* level_z;    /* [N] the standard normal behind each arena's per-level command re-draw (KAMMA.py:539, :561, :566) */
  const int32_t* levels;   /* [N] terrain level per arena for that re-draw (self.terrain_levels without a terrain attached) */
} orc_test_hooks;
int orc_test_set_hooks(orc_env* env, const orc_test_hooks* hooks);

/* host pointer to a buffer with the SAME id / shape / dtype as ois_buffer() */
int orc_buffer(orc_env* env, int id, void** ptr, size_t* bytes);
int orc_episode_info(orc_env* env);               /* fills OIS_BUF_EPISODE_INFO */
int orc_stats_fetch(orc_env* env, ois_stats* out, int reset_after);
const char* orc_last_error(void);
/* FIFO transition memory mirror (ois_replay_*); memory_size 0 detaches */
int orc_replay_attach(orc_env* env, int memory_size);
int orc_replay_buffer(orc_env* env, int id, void** ptr, size_t* bytes);
int orc_replay_cursor(const orc_env* env, int* memory_index, int* filled);

/* height-field terrain mirror (ois_terrain_attach / ois_height_scan); all pointers are host memory */
int orc_terrain_attach(orc_env* env, const ois_terrain* terrain, const int16_t* heights,
                       const float* terrain_origins, const int32_t* levels_types);
int orc_height_scan(orc_env* env, float* heights_out /*[N,140]*/);
/* get_heights (KAMMA.py:872-893) on explicit world poses: pos3 [n,3], quat4 [n,4] (w,x,y,z) */
void orc_get_heights(int n, const float* pos3, const float* quat4, const int16_t* heights, int rows, int cols,
                     float horizontal_scale, float vertical_scale, float border_size, float* out);

/* mirror of ois_gae on host tensors [T, N] (ppo.py:345-358 / ppoc.py:325-336) */
int orc_gae(int mode, int T, int N, const float* rewards, const uint8_t* terminated, const float* values,
            const float* last_values, double gamma, double lambda, int normalize, float* returns_out,
            float* advantages_out);

/* ---- unit-level API: the reference-followed formulas, one call per SURVEY §8a row,
 *      array-in / array-out so the golden fixtures can be replayed ---- */

/* a2: criteria controller, KAMMA.py:604-624 */
void orc_criteria(int n, const float* cmd_x, const float* cmd_yaw, const float* v_fwd_prev,
                  const float* omega_z, float Kp, float Kp_omega, float mass, float inertia,
                  float dt, float* v_delta, float* omega_delta, float* ac_left, float* ac_right);
/* a3: gamma schedule, KAMMA.py:627-628 */
void orc_gamma(int n, const int32_t* sim_steps, float max_sim_steps, float* gamma);
/* a4: stochastic switch KAMMA.py:632-636 (rand given) / blend KA_DDPG_4D.py:627-631;
 *     a5 clip KAMMA.py:649; guiding reward KAMMA.py:656 / KA_DDPG_4D.py:637 */
void orc_mix(int n, int mix_mode, int guiding_mode, const float* actions4, const float* ac_left,
             const float* ac_right, const float* gamma1, const float* rand01, float action_scale,
             float clip, float* torques4, float* wheel_torqs4, float* guiding);
/* a7: is_done, KAMMA.py:736-764 */
void orc_is_done(int n, const int32_t* episode_buf, int max_episode_length, const float* quat4,
                 const float* pos3, float bounds, uint8_t* reset, uint8_t* timeout);
/* a8: calculate_metrics, KAMMA.py:767-822. sums8 is updated in place, prev_lin/prev_ang too */
void orc_metrics(int n, const float* v_fwd_prev, const float* omega_z, float* prev_lin,
                 float* prev_ang, const float* v_delta, const float* omega_delta,
                 const float* wheel_torqs4, const float* guiding, const float* gamma2,
                 const uint8_t* reset, const uint8_t* timeout, float dt, float w1, float w2,
                 float w3, float sparse_bonus, float sparse_thr, float terminal_reward,
                 float* lin_acc, float* ang_acc, float* rew, float* sums8);
/* a11: forward-velocity projection, KAMMA.py:701-710 */
void orc_eval_sums(int n, const float* old4, const float* wt4, const float* desired_v, const float* v_delta,
                   float* sums_c4); /* KAMMA_eval.py:677-683, :840 */
void orc_forward_projection(int n, const float* quat4, const float* linvel3, float* v_fwd);
/* quat helpers as used by the reference (omni.isaac.core.utils.torch.rotations, wxyz) */
void orc_quat_apply(const float q[4], const float v[3], float out[3]);
void orc_quat_mul(const float a[4], const float b[4], float out[4]);
/* a9: the pose/command part of reset_idx for one arena given its 8 uniform draws
 *     (order: pos x, pos y, yaw, cmd x, cmd y, cmd heading, x re-draw, discarded omega) */
void orc_reset_draws(const ois_config* cfg, const float u[8], float pos3[3], float quat4[4],
                     float commands4[4]);

/* a9: what one resetting arena contributes to extras["episode"] (KAMMA.py:470-485); v: the first 16 OIS_EI_* slots */
void orc_reset_ratios(const float S[8], float gamma2, float episode_length_s, float v[16]);

/* RNG + deterministic trig */
void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
float orc_u01(uint32_t x);
void orc_sincos(float x, float* s, float* c);

/* one physics substep for one arena (docs/SPEC.md §3); exposed for physics unit tests.
 * state: p[3], q[4], v[3], w[3] (world), spin[4] (absolute wheel spin about body y),
 * lam_*[4]: accumulated normal / longitudinal / lateral contact impulses (warm start) */
void orc_substep(const ois_config* cfg, float mu, const float n_world[3], const float tau[4],
                 float p[3], float q[4], float v[3], float w[3], float spin[4], float lam_n[4],
                 float lam_l[4], float lam_t[4]);

int orc_push_robots(orc_env* env);

/* IEEE binary16 <-> binary32, round to nearest even (the storage format of OIS_BUF_LAMBDA_*) */
uint16_t orc_f2h(float f);
float orc_h2f(uint16_t h);

/* the per-terrain-level command generators (KAMMA.py:533-567) and the deterministic helpers they use */
float orc_sin(float x);
float orc_normal(float u1, float u2);
void orc_box_bounds(int32_t sim_steps, float max_sim_steps, float lo, float hi, float* xmin, float* xmax); /* KAMMA_curr1.py:579-591 */
float orc_grid_command(int32_t sim_steps, float max_sim_steps, float lo, float hi, float u_side, float z); /* KAMMA_curr2.py:602-626 */
float orc_eval_command(int32_t sim_steps, float ramp_steps); /* KAMMA_eval.py:545-550 */
void orc_level_command(int level, float fraction, int32_t step_counter, float z, double dt, double episode_length_s,
                       float* cmd_x, float* cmd_yaw);

#ifdef __cplusplus
}
#endif
#endif
