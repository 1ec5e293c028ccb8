/*
 * ois.h — C ABI of the MI355X-native vectorised skid-steer torque-control environment.
 *
 * This is the drop-in boundary for ONE hot path of meesjansen/Options_In_Simulation:
 * the reset()/step() loop of `TorqueDistributionTask` (reference: src/my_envs/KAMMA.py
 * and its 11 copy-paste variants) for N independent arenas.
 * The reference has no FFI of its own (it is pure Python on top of Isaac Sim/PhysX); the
 * entry points below are what a ctypes binding of that task would bind, one per
 * reference call the host loop makes:
 *
 *   ois_create / ois_destroy    <- TorqueDistributionTask.__init__ + update_config
 *                                  (KAMMA.py:149-228, 231-275), RLTask buffers
 *                                  (rl_task.py:145-154)
 *   ois_reset                   <- post_reset -> reset_idx(arange(N))   (KAMMA.py:366-414)
 *   ois_step                    <- VecEnv.step = pre_physics_step + controlFrequencyInv
 *                                  world steps + post_physics_step
 *                                  (KAMMA.py:596-656, 43, 674-727)
 *   ois_rollout                 <- K consecutive ois_step calls fused into one launch
 *                                  (no reference counterpart; BASELINE.json configs 1/2)
 *   ois_buffer                  <- obs_buf / rew_buf / reset_buf / timeout_buf /
 *                                  wheel_torqs / commands / base_* tensors
 *   ois_outputs_copy            <- the wrapper's per-step clones of obs / reward / terminated + zeros(truncated)
 *   ois_episode_info            <- extras["episode"] filled in reset_idx (KAMMA.py:480-488)
 *   ois_env0_diag               <- task.observed_components (KAMMA.py:840-867)
 *   ois_logs_enable / _fetch    <- both of the above as the trainer reads them: after EVERY step, on the
 *                                  host (src/my_trainers/base_KA_DDPG.py:184-212, 259-262)
 *   ois_stats_fetch             <- finished-episode return tracking of the agent base
 *                                  class (src/my_agents/base.py:305-341)
 *   ois_replay_*                <- FIFOMemory.add_samples as fed by record_transition
 *                                  (train_kaddpg_4d_nofifo_random_RLIL.py:105-206,
 *                                  base_KA_DDPG.py:205-224): the step kernel writes the row
 *   ois_gae                     <- compute_gae (src/my_agents/ppo.py:321-360) and
 *                                  compute_advantages_and_returns (ppoc.py:305-336)
 *   ois_terrain_attach          <- get_terrain / _create_trimesh / update_SI_level
 *                                  (KAMMA.py:296-349, 502-529; commented out there)
 *   ois_height_scan             <- get_heights (KAMMA.py:872-893)
 *   ois_push_robots             <- push_robots (KAMMA.py:729-733; defined, never called there)
 *   ois_randomize_params, ois_save_state / ois_load_state: no reference counterpart
 *                                  (BASELINE.json configs[3]; SURVEY.md §5 checkpoint)
 *
 * Conventions
 *   - plain C, no torch types; every pointer named *device* is a HIP device pointer.
 *   - the library never allocates device memory: the caller hands one workspace of
 *     ois_workspace_bytes() bytes (hipMalloc or a torch uint8 tensor) and every
 *     buffer returned by ois_buffer() aliases into it (zero copy).
 *   - every call returns 0 (OIS_OK) or a negative OIS_E* code; the message is in
 *     ois_last_error() (thread local). No C++ exception crosses the boundary.
 *   - all GPU work is enqueued on the caller's hipStream_t (`stream`, may be NULL for
 *     the default stream); only the *_fetch / host-output calls synchronise that stream.
 *   - one ois_env is owned by one host thread at a time (no internal locking).
 */
#ifndef OIS_H_
#define OIS_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2: ois_replay_*, ois_gae, ois_terrain_attach, ois_height_scan, buffers 21-22, a larger workspace
 * (state blobs of version 1 do not load)
 * 3: ois_terrain_attach takes the min-field scratch (one gather per height sample)
 * 4: ois_config: wheel_limit_velocity / wheel_limit_damping (two of the reserved words); state blobs of
 *    version 3 do not load (the solver and the joint model changed, see docs/SPEC.md 3.3, 3.5);
 *    OIS_CMD_LEVEL, OIS_BUF_CMD_FRACTION, ois_profile_* */
/* 5: ois_kernel_info; state blobs carry the flat-plane flag (version 4 blobs do not load); ois_config validation
 *    also refuses an unstable joint resistance and a negative depenetration velocity */
/* 6: ois_logs_enable / ois_logs_fetch (the step's logs through pinned host memory); ois_episode_info and ois_stats_fetch are
 *    restructured (one launch of one workgroup / two launches); the workspace grew by one 256-byte slot (state blobs of
 *    version 5 do not load); ois_terrain grew by footprint_stride (struct_bytes = 52) */
#define OIS_ABI_VERSION 6

/* error codes */
#define OIS_OK 0
#define OIS_EINVAL (-1)    /* bad argument / config */
#define OIS_ENOMEM (-2)    /* workspace too small or host allocation failed */
#define OIS_EHIP (-3)      /* a HIP runtime call failed (message has hipGetErrorString) */
#define OIS_ESTATE (-4)    /* call order violated (e.g. step before reset) */

/* mix_mode: how the policy action and the expert ("criteria") action are combined */
#define OIS_MIX_SWITCH 0   /* KAMMA.py:632-635      mask = U[0,1) < gamma1 */
#define OIS_MIX_BLEND 1    /* KA_DDPG_4D.py:627-631 (1-g)*a*scale + g*criteria */

/* gamma modes */
#define OIS_GAMMA_SCHEDULE 0 /* clamp(1 - sim_steps/max_sim_steps, 0)  KAMMA.py:627-628 */
#define OIS_GAMMA_CONST 1    /* commented-out schedule lines: KAMMA_g11_g21.py:627-628,
                                KAMMA_g1d_g21.py:628, KA_DDPG_2D.py:627; eval: 0 */

/* guiding_mode: which torque the guiding reward compares with the criteria action */
#define OIS_GUIDE_EXECUTED 0 /* clipped executed torque   KAMMA.py:656 */
#define OIS_GUIDE_POLICY 1   /* unclipped a*action_scale  KA_DDPG_4D.py:637 */

/* cmd_sampler: how commands[:,0] is drawn at reset (sample_velocity_command) */
#define OIS_CMD_RAND 0 /* KAMMA.py:569-575 */
#define OIS_CMD_BOX 1  /* KAMMA_curr1.py:577-598 */
#define OIS_CMD_GRID 2 /* KAMMA_curr2.py:602-640 */
#define OIS_CMD_EVAL 3 /* KAMMA_eval.py:545-550 */
#define OIS_CMD_LEVEL 4 /* terrain.curriculum: the per-terrain-level generators of sample_velocity_command
                           (KAMMA.py:533-567), re-drawn for EVERY arena after EVERY step (KAMMA.py:719-725) and
                           not at a reset (KAMMA.py:490). Also switched on by ois_terrain_attach with
                           curriculum != 0. Costs one small extra launch per policy step. Not with OIS_CMD_EVAL:
                           the *_eval files test evalsampling first (KAMMA_eval.py:544); attaching such a terrain to an
                           EVAL environment is OIS_EINVAL. */

/* rollout policies */
#define OIS_POLICY_ZERO 0     /* actions == 0                                  */
#define OIS_POLICY_RANDOM 1   /* actions ~ U[0,1)^A from Philox(seed, arena, step) */
#define OIS_POLICY_EXTERNAL 2 /* actions[k][N][A] supplied by the caller       */

typedef struct ois_config {
  uint32_t struct_bytes; /* = sizeof(ois_config); checked by ois_create */
  int32_t num_envs;      /* arenas owned by this process (this GPU)     */
  int64_t arena_offset;  /* global id of local arena 0 (RNG key; sharding invariant) */
  uint64_t seed;         /* TASK_CFG["seed"] (KAMMA.py:35) */

  /* ---- variant switches ---- */
  int32_t action_dim;   /* 4 (KAMMA, KA_DDPG_4D) | 2 (KA_DDPG_2D.py:623-624) | 1 (KA_DDPG_1D.py:600-603) */
  int32_t mix_mode;     /* OIS_MIX_*   */
  int32_t gamma1_mode;  /* OIS_GAMMA_* action assist */
  int32_t gamma2_mode;  /* OIS_GAMMA_* reward assist */
  float gamma1_const;
  float gamma2_const;
  int32_t guiding_mode; /* OIS_GUIDE_* */
  int32_t cmd_sampler;  /* OIS_CMD_*   */
  float max_sim_steps;  /* KAMMA.py:179 (400000), _medium 500000, KA_DDPG_4D 700000 */
  int32_t eval_sums;    /* 1: KAMMA_eval extra episode sums (KAMMA_eval.py:677-685) */

  /* ---- task constants (KAMMA.py update_config + hard-coded) ---- */
  float sim_dt;               /* TASK_CFG sim.dt 0.01              KAMMA.py:82  */
  int32_t decimation;         /* control.decimation 10             KAMMA.py:75  */
  int32_t extra_substeps;     /* env.controlFrequencyInv 10        KAMMA.py:43 (loop body is third-party) */
  float policy_dt;            /* self.dt = decimation * sim.dt = 0.1 (float32 of the Python double) KAMMA.py:260 */
  int32_t max_episode_length; /* int(episodeLength_s / dt + 0.5) = 100   KAMMA.py:262 */
  float episode_length_s;     /* learn.episodeLength_s 10          KAMMA.py:70  */
  float Kp;                   /* control.stiffness (0.035 from the train scripts) KAMMA.py:263 */
  float Kp_omega;             /* 0.665                             KAMMA.py:613 */
  float vehicle_mass;         /* 25 (controller constant)          KAMMA.py:174 */
  float vehicle_inertia;      /* 1.05 (controller constant)        KAMMA.py:175 */
  float action_scale;         /* control.actionScale 3.0           KAMMA.py:78  */
  float torque_clip;          /* 4.0                               KAMMA.py:649 */
  float w1, w2, w3;           /* -25, -0.02, -0.09                 KAMMA.py:785 */
  float sparse_bonus;         /* 0.2                               KAMMA.py:792 */
  float sparse_threshold;     /* 0.01                              KAMMA.py:790-791 */
  float terminal_reward;      /* learn.terminalReward 0.0          KAMMA.py:69, 801 */
  float cmd_x_lo, cmd_x_hi;   /* randomCommandVelocityRanges.linear_x  KAMMA.py:71 */
  float cmd_y_lo, cmd_y_hi;   /* linear_y                          KAMMA.py:72  */
  float cmd_yaw_lo, cmd_yaw_hi; /* yaw                             KAMMA.py:73  */
  float cmd_x_span, cmd_y_span, cmd_yaw_span; /* float32 of the Python DOUBLE (hi - lo): torch_rand_float
                                 multiplies the fp32 draw by that scalar (KAMMA.py:448-456) */
  float init_pos[3];          /* baseInitState.pos                 KAMMA.py:44  */
  float init_quat[4];         /* baseInitState.rot (w,x,y,z)       KAMMA.py:45  */
  float init_lin_vel[3];      /* vLinear                           KAMMA.py:46  */
  float init_ang_vel[3];      /* vAngular                          KAMMA.py:47  */
  float init_dof_vel[4];      /* dofInitVelocities                 KAMMA.py:50  */
  float spawn_xy_range;       /* 0.5: U(-0.5,0.5)                  KAMMA.py:424 */
  float bounds;               /* 50                                KAMMA.py:187 */
  float clip_obs;             /* env.clipObservations 1000         KAMMA.py:42, rl_task.py:99 */
  float eval_ramp_steps;      /* 50000                             KAMMA_eval.py:545-550 */

  /* ---- physics model (build-defined, parity-unpinned: PhysX is not in the reference;
          parameters from src/my_assets/urdf/origin_sym.urdf, origin_v18.py:61-70,
          KAMMA.py:82-127; model in docs/SPEC.md §3) ---- */
  float gravity;              /* 9.81 (sim.gravity z)              KAMMA.py:84  */
  float body_mass;            /* 21 + 4*1 = 25                     urdf:11,34   */
  float body_inertia[3];      /* composite about composite COM, wheel spin excluded (host-composed) */
  float com_height;           /* composite COM above base-link origin (0.00896)  */
  float wheel_pos[3];         /* |x|,|y|,z of wheel centres in base-link frame   urdf:115-139 */
  float wheel_radius;         /* 0.1175                            urdf:40, KAMMA.py:79 */
  float wheel_spin_inertia;   /* 0.01                              urdf:35      */
  float wheel_damping;        /* viscous joint damping c1 [N m s/rad], CALIBRATED (0: tests/golden/ref_stats.json) */
  float friction;             /* default mu (0.85 from scripts)    train_*.py:66-67 */
  float baumgarte;            /* penetration bias factor                          */
  float max_depenetration_velocity; /* 100                         KAMMA.py:105 */
  float max_angular_velocity; /* 64 rad/s                          origin_v18.py:70. (Linear speed has no cap; an arena whose
                                 speed exceeds 50 m/s after a policy step's substeps, or whose state is not finite, is discarded
                                 and reset as fallen: the safety net of docs/SPEC.md 3.9) */
  int32_t solver_iterations;  /* 4                                 KAMMA.py:96-97 */
  float wheel_limit_velocity; /* joint velocity limit [rad/s]: <limit velocity="10"/> urdf:117,125,133,141 */
  float wheel_limit_damping;  /* resistance above the limit [N m s/rad], CALIBRATED with wheel_damping
                                 (docs/SPEC.md section 3.3, tests/golden/ref_stats.json); 0 = no limit */
  float curriculum_threshold_high; /* 10.0: self.threshold_high, the divisor of `fraction` (KAMMA.py:269, 498) */
  int32_t env0_reward_guard;  /* 1: KAMMA_g1d_g21.py:838-842 — get_observations overwrites rew_buf[0] with 1e-12 when
                                 |rew_buf[0]| < 1e-12 (a division guard for the env-0 percentages), BEFORE the reward is
                                 returned: the returned reward, the diagnostics and the transition row of LOCAL arena 0
                                 carry it, the episode sums (added in calculate_metrics) do not */
  int32_t reserved_[4];
} ois_config;

/* Buffer ids for ois_buffer(). Shapes are row-major; N = num_envs. */
enum ois_buffer_id {
  OIS_BUF_OBS = 0,         /* float [N,6]  obs_buf            KAMMA.py:833 (clamped, rl_task.py:99) */
  OIS_BUF_REW = 1,         /* float [N]    rew_buf            KAMMA.py:798-801 */
  OIS_BUF_DONE = 2,        /* uint8 [N]    reset_buf (0/1)    KAMMA.py:754-760 */
  OIS_BUF_TIMEOUT = 3,     /* uint8 [N]    timeout_buf        KAMMA.py:740-744 */
  OIS_BUF_WHEEL_TORQS = 4, /* float [N,4]  wheel_torqs        KAMMA.py:649 */
  OIS_BUF_POSE = 5,        /* float [N,4]  p_com.xyz, v_forward_projected */
  OIS_BUF_QUAT = 6,        /* float [N,4]  base_quat (w,x,y,z) */
  OIS_BUF_LINVEL = 7,      /* float [N,4]  v.xyz (world), previous_linear_velocity[:,0] */
  OIS_BUF_ANGVEL = 8,      /* float [N,4]  w.xyz (world), previous_angular_velocity[:,2] */
  OIS_BUF_WHEEL_VEL = 9,   /* float [N,4]  dof_vel (joint velocities, order fl,rl,fr,rr) */
  OIS_BUF_COMMANDS = 10,   /* float [N,4]  commands (x, y, yaw, heading)  KAMMA.py:378-380 */
  OIS_BUF_COUNTERS = 11,   /* int32 [N,4]  episode_buf, sim_steps, episode_count, reserved */
  OIS_BUF_SUMS_A = 12,     /* float [N,4]  episode sums r1, r2, r3, dense */
  OIS_BUF_SUMS_B = 13,     /* float [N,4]  episode sums sparse, guiding, observed, final */
  OIS_BUF_PARAMS = 14,     /* float [N,4]  per-arena mu, plane normal xyz (domain randomisation). mu may be written in place;
                              after writing NORMALS into it yourself call ois_params_dirty (the flat-plane kernels ignore them) */
  OIS_BUF_EPISODE_INFO = 15, /* float [OIS_EPISODE_INFO_LEN] written by ois_episode_info */
  OIS_BUF_ENV0_DIAG = 16,  /* float [OIS_ENV0_DIAG_LEN] arena-0 diagnostics of the last step */
  OIS_BUF_SUMS_C = 17,     /* float [N,4]  eval sums smoothness, tracking error, desired v, prev-torque scratch */
  OIS_BUF_LAMBDA_N = 18,   /* half [N,4]   contact impulses carried from one policy step to the next as the solver's warm
                              start: normal, per wheel. IEEE binary16, rounded to nearest even at the end of every policy
                              step (docs/SPEC.md 3.5): a warm start needs no more, and the three rows cost 48 B per
                              arena-step instead of 96 */
  OIS_BUF_LAMBDA_L = 19,   /* half [N,4]   longitudinal friction impulses */
  OIS_BUF_LAMBDA_T = 20,   /* half [N,4]   lateral friction impulses */
  OIS_BUF_ENV_ORIGINS = 21, /* float [N,4] env_origins xyz of the arena's terrain tile (KAMMA.py:397, 529); 0 = no terrain */
  OIS_BUF_TERRAIN_IDX = 22, /* int32 [N,4] terrain_levels, terrain_types (KAMMA.py:342-347), -, - */
  OIS_BUF_CMD_FRACTION = 23, /* float [N]  self.fraction: r1 episode sum / threshold_high as of the last step in which
                                 ANY arena of this environment was reset (KAMMA.py:498); OIS_CMD_LEVEL only */
  OIS_BUF_COUNT_
};

#define OIS_EPISODE_INFO_LEN 32
#define OIS_ENV0_DIAG_LEN 32

/* index of each value inside OIS_BUF_EPISODE_INFO (names: KAMMA.py:205-221, 470-488) */
enum ois_episode_info_slot {
  OIS_EI_COUNT = 0,        /* number of arenas reset in the step that produced this record */
  OIS_EI_R1 = 1, OIS_EI_R2 = 2, OIS_EI_R3 = 3, OIS_EI_DENSE = 4, OIS_EI_SPARSE = 5,
  OIS_EI_GUIDING = 6, OIS_EI_OBSERVED = 7, OIS_EI_FINAL = 8,
  OIS_EI_R1_OVER_FINAL = 9, OIS_EI_R2_OVER_FINAL = 10, OIS_EI_R3_OVER_FINAL = 11,
  OIS_EI_DENSE_OVER_FINAL = 12, OIS_EI_SPARSE_OVER_FINAL = 13,
  OIS_EI_GUIDING_OVER_FINAL = 14, OIS_EI_OBSERVED_OVER_FINAL = 15,
  OIS_EI_GAMMA1 = 16, OIS_EI_GAMMA2 = 17, OIS_EI_TERRAIN_LEVEL = 18,
  OIS_EI_STEP = 19,        /* sim_steps value at which the record was produced (as float) */
  OIS_EI_SMOOTHNESS = 20, OIS_EI_TRACKING_ERROR = 21, OIS_EI_DESIRED_VELOCITY = 22
};

/* finished-episode statistics since the last ois_stats_fetch(reset_after=1)
 * (device-side replacement of src/my_agents/base.py:305-341) */
typedef struct ois_stats {
  double episodes;      /* finished episodes                       */
  double return_sum;    /* sum of episode returns ("Final reward") */
  double return_sqsum;  /* sum of squared returns                  */
  double return_min;
  double return_max;
  double length_sum;    /* sum of episode lengths in policy steps  */
  double steps;         /* env steps executed (N * policy steps)   */
  double reserved;
} ois_stats;

typedef struct ois_env ois_env; /* opaque */

int ois_abi_version(void);
const char* ois_last_error(void);

/* Fill `cfg` with the reference defaults: TASK_CFG of KAMMA.py:29-128 with the overrides
 * every train script applies (train_kamma_4d_nofifo_random_RLIL.py:63-91: stiffness 0.035,
 * friction 0.85, linear_x [1,2]) and the KAMMA_medium schedule (max_sim_steps 500000). */
int ois_config_default(ois_config* cfg);

/* Bytes of device workspace ois_create needs for `cfg` (0 on error). */
size_t ois_workspace_bytes(const ois_config* cfg);

/* Bind an environment to a caller-owned device workspace (>= ois_workspace_bytes, 256-B
 * aligned). `device` is the HIP device ordinal the workspace lives on. */
int ois_create(const ois_config* cfg, int device, void* workspace_device, size_t workspace_bytes,
               ois_env** out);
void ois_destroy(ois_env* env);

/* post_reset: zero all task buffers, reset_idx(all arenas) with `seed`. */
int ois_reset(ois_env* env, uint64_t seed, void* stream);

/* One policy step for all arenas. actions_device: float [N, action_dim] row-major, aligned to
 * action_dim * 4 bytes (the kernel reads an arena's actions with one vector load; OIS_EINVAL otherwise).
 * Every entry point makes the env's device current for the call and restores the caller's. */
int ois_step(ois_env* env, const float* actions_device, void* stream);

/* k policy steps in ONE launch; state stays in registers between steps.
 * policy EXTERNAL: actions_device is float [k, N, action_dim]; otherwise ignored (may be NULL).
 * reward_sum_device (optional, may be NULL): float [N], receives the sum of the k rewards. */
int ois_rollout(ois_env* env, int k, int policy, const float* actions_device,
                float* reward_sum_device, void* stream);

/* Kernel-only timing of the step launches (measurement aid, no reference counterpart). Between
 * begin and end every ois_step / ois_rollout launch carries a start and a stop event on its own
 * dispatch packet (hipExtLaunchKernel), so the elapsed time is the kernel's begin -> end as
 * rocprofv3 --kernel-trace reports it: no launch gap, no marker packets between kernels. end
 * synchronises with the last profiled launch and returns the summed kernel time and the launch
 * count. Not for use under HIP-graph capture (OIS_ESTATE), like a step with a transition memory attached:
 * both keep a host-side cursor that advances per call, not per replay. */
int ois_profile_begin(ois_env* env);
int ois_profile_end(ois_env* env, double* kernel_ms_total, long long* launches);

/* Device pointer + size of one buffer (aliases the workspace). */
int ois_buffer(ois_env* env, int id, void** ptr_device, size_t* bytes);

/* The step's return values as arrays of the CALLER's: the wrapper between the task and the trainer returns fresh tensors from
 * every step() (clones of obs_buf / rew_buf / reset_buf and a new all-false `truncated`; skrl's wrapper, third-party; call site
 * src/my_trainers/base_KA_DDPG.py:182). One launch: obs_out float [N,6], rew_out float [N], terminated_out / truncated_out
 * uint8 [N] (truncated is filled with 0); any of them may be NULL. Device memory, obs_out 8-byte and rew_out 4-byte aligned. */
int ois_outputs_copy(ois_env* env, float* obs_out, float* rew_out, uint8_t* terminated_out, uint8_t* truncated_out,
                     void* stream);

/* Per-arena domain randomisation (BASELINE.json config 4): mu ~ U(mu_lo, mu_hi) and plane
 * slope drawn from `slopes[n_slopes]` (tan of the incline about a random heading), both
 * from Philox keyed by the global arena id. n_slopes == 0 keeps the flat plane. 0 <= mu_lo <= mu_hi (finite), at most 16
 * slopes with |slope| <= 10 (OIS_EINVAL otherwise; ois_config.friction likewise >= 0). The plane passes through the arena's origin;
 * a reset (ois_reset, and every reset_idx inside a step) sets the robot down ON it, aligned with it (docs/SPEC.md 3.2) — so
 * call this BEFORE ois_reset: a robot already standing keeps its pose and meets the new plane where it is until its next reset. */
int ois_randomize_params(ois_env* env, float mu_lo, float mu_hi, const float* slopes_host,
                         int n_slopes, void* stream);

/* The caller wrote plane normals into OIS_BUF_PARAMS itself (its own domain randomisation): the environment leaves the
 * flat-plane kernels, which take n = (0, 0, 1) without reading it (ois_kernel_info: plane=flat), for the general ones — until
 * the next ois_randomize_params without slopes or ois_load_state of a flat blob. Friction (column 0) needs no such call. */
int ois_params_dirty(ois_env* env);

/* Which step-kernel instantiation ois_step / ois_rollout launch for this environment right now (measurement aid:
 * bench.py keys its per-kernel figures on it), e.g. "step_kernel<1,1,4,1> lanes_per_arena=1 packed iterations=4(unrolled)
 * plane=flat" (the template arguments: lanes per arena, packed, unrolled iterations, flat). plane=flat: no ois_randomize_params slopes since ois_create (or a blob saved so) and no terrain — the
 * kernel then takes the plane normal as (0, 0, 1) and does not read it from OIS_BUF_PARAMS; a caller that writes
 * normals into that buffer itself says so with ois_params_dirty (or sets OIS_FLAT=0). */
int ois_kernel_info(const ois_env* env, char* out, size_t cap);

/* push_robots (KAMMA.py:729-733): the world linear velocity x, y of EVERY arena is replaced by U(-1, 1)
 * draws (Philox purpose 7, counter = the arena's sim_steps; torch_rand_float(-1.0, 1.0, ...)). The reference
 * defines it and never calls it; here for scripts that do. */
int ois_push_robots(ois_env* env, void* stream);

/* Reduce the per-group records of the most recent reset event into OIS_BUF_EPISODE_INFO (device; one launch up to
 * 131 072 arenas, two above or on a terrain with a level curriculum). No host synchronisation. A caller that rewrites
 * OIS_BUF_TERRAIN_IDX itself re-attaches the terrain (the level mean of a terrain without curriculum is taken at attach). */
int ois_episode_info(ois_env* env, void* stream);

/* Copy finished-episode statistics to the host (synchronises `stream`). */
int ois_stats_fetch(ois_env* env, ois_stats* host_out, int reset_after, void* stream);

/* Copy the arena-0 diagnostics block to the host (synchronises `stream`). */
int ois_env0_diag(ois_env* env, float* host_out /*[OIS_ENV0_DIAG_LEN]*/, void* stream);

/* ---- the step's logs on the host, for a caller that reads them after every step ----
 * The reference's trainer reads, after EVERY env.step(), the 26 values of task.observed_components and every value of
 * infos["episode"] with .item() (src/my_trainers/base_KA_DDPG.py:184-212, 259-262; producers KAMMA.py:480-488, 840-867).
 * Through ois_episode_info + a copy that is two launches, a blocking transfer and a drained stream per step.
 *   ois_logs_fetch: the [OIS_EPISODE_INFO_LEN] extras["episode"] values (slots: ois_episode_info_slot) and the
 *     [OIS_ENV0_DIAG_LEN] arena-0 diagnostics of the CURRENT state into two host arrays. If no finalisation of the current
 *     state is in flight it enqueues one on `stream` (one launch — two above 131 072 arenas — which also fills OIS_BUF_EPISODE_INFO); then it waits
 *     until the kernel has published the values: the kernel writes them into a pinned, host-coherent block the library
 *     allocates on first use (hipHostMalloc, 4 KiB — host memory; device memory stays the caller's) and stores a sequence
 *     number behind them, the host polls that word. No stream synchronisation, no copy engine; work enqueued behind the
 *     finalisation keeps running while the host reads.
 *   ois_logs_enable(env, 1): from now on ois_step enqueues that finalisation itself, right behind the step kernel on the
 *     step's stream, so that the fetch finds the values already published (or on their way) instead of starting work on a
 *     drained stream. 0 switches it off again (a loop that does not read logs saves the launch). Not done for a step under
 *     HIP-graph capture (the sequence number advances per call, not per replay); ois_rollout never does it.
 * The fetch must be given the stream the finalisation was (or is to be) enqueued on. OIS_ESTATE under stream capture. */
int ois_logs_enable(ois_env* env, int on);
int ois_logs_fetch(ois_env* env, float* episode_info_host /*[OIS_EPISODE_INFO_LEN]*/,
                   float* env0_diag_host /*[OIS_ENV0_DIAG_LEN]*/, void* stream);

/* ---- device-resident FIFO transition memory written by the step kernel ----
 * Semantics of the train scripts' FIFOMemory (train_kaddpg_4d_nofifo_random_RLIL.py:105-206:
 * tensors [memory_size, num_envs, dim], row memory_index overwritten, index wraps) with the
 * trainer's executed-action substitution (src/my_trainers/base_KA_DDPG.py:212:
 * actions = task.wheel_torqs / task.action_scale). Once attached, every policy step of
 * ois_step / ois_rollout appends one row: states = the observation before the step,
 * actions = wheel_torqs / action_scale [4], rewards, next_states = the observation after the
 * step, terminated. The memory is caller-owned device memory, like the workspace. */
enum ois_replay_id {
  OIS_RP_STATES = 0,      /* float [M, N, 6] */
  OIS_RP_ACTIONS = 1,     /* float [M, N, 4] */
  OIS_RP_REWARDS = 2,     /* float [M, N, 1] */
  OIS_RP_NEXT_STATES = 3, /* float [M, N, 6] */
  OIS_RP_TERMINATED = 4,  /* uint8 [M, N, 1] */
  OIS_RP_COUNT_
};
size_t ois_replay_bytes(const ois_env* env, int memory_size);
/* attach (memory_size >= 1) or detach (memory_device == NULL). Resets the cursor. The library
 * never clears the memory: rows not yet written keep what the caller put there (skrl's
 * tensors start as zeros, so the host mirror hands over a zero-filled block). */
int ois_replay_attach(ois_env* env, int memory_size, void* memory_device, size_t bytes);
int ois_replay_buffer(ois_env* env, int id, void** ptr_device, size_t* bytes);
/* memory_index = next row to be written, filled = 1 once the index has wrapped */
int ois_replay_cursor(const ois_env* env, int* memory_index, int* filled);
/* Gather a mini-batch in ONE launch (FIFOMemory.sample / sample_by_index,
 * train_kaddpg_4d_nofifo_random_RLIL.py:112-145: five indexed copies there). indexes_device:
 * int64 [count] positions in the flattened [memory_size * num_envs] sample space (values outside
 * it are clamped into it). Outputs are contiguous device arrays [count, 6|4|1|6|1]; any of them
 * may be NULL to skip that tensor. */
int ois_replay_sample(ois_env* env, const int64_t* indexes_device, int count, float* states_out,
                      float* actions_out, float* rewards_out, float* next_states_out,
                      uint8_t* terminated_out, void* stream);

/* ---- generalized advantage estimation over a rollout ----
 * The backward recurrence the on-policy agents run in a Python loop over the rollout memory,
 * one thread per arena here. All pointers are device memory, row-major [T, N] (the memory
 * tensors [T, N, 1] as they are), terminated is uint8/bool, last_values is [N].
 *   OIS_GAE_PPO  (src/my_agents/ppo.py:345-358):
 *       adv_t = r_t - v_t + gamma*(1-term_t) * (v_{t+1} + lambda*adv_{t+1}),  v_T = last_values
 *       returns = adv + values; then, if normalize, adv = (adv - mean(adv)) / (std(adv) + 1e-8)
 *       with the unbiased std over all T*N entries (torch.std).
 *   OIS_GAE_PPOC (src/my_agents/ppoc.py:325-336):
 *       delta_t = r_t + gamma*(1-term_t)*v_{t+1} - v_t;  adv_t = delta_t + (gamma*lambda)*(1-term_t)*adv_{t+1}
 *       returns = adv + values; the reference never normalises here (normalize may still be set).
 * Each float operation is rounded separately, in the order the reference's expression
 * evaluates, so the un-normalised outputs are bit-identical to the torch loop; gamma and lambda
 * are the Python doubles (torch rounds a scalar operand to float32 when it meets a float32
 * tensor: ppo uses f32(gamma) and f32(lambda), ppoc f32(gamma) and f32(gamma*lambda)).
 * scratch_device: ois_gae_scratch_bytes(N) bytes, only needed when normalize != 0.
 * ois_gae has no environment handle: it launches on the CALLER's current device, which must be the one the
 * pointers belong to (the Python wrapper holds a torch.cuda.device guard around the call). */
enum ois_gae_mode { OIS_GAE_PPO = 0, OIS_GAE_PPOC = 1 };
size_t ois_gae_scratch_bytes(int num_envs);
int ois_gae(int mode, int T, int num_envs, const float* rewards, const uint8_t* terminated,
            const float* values, const float* last_values, double gamma, double lambda, int normalize,
            float* returns_out, float* advantages_out, void* scratch_device, size_t scratch_bytes,
            void* stream);

/* ---- height-field terrain ----
 * The reference builds ONE int16 height field for all robots (my_utils/origin_terrain_generator.py
 * Terrain: tiles of mapLength x mapWidth metres, numLevels x numTerrains of them, a border
 * around; height = raw * vertical_scale, pixel = horizontal_scale) and places robot i on the
 * tile terrain_origins[terrain_levels[i], terrain_types[i]] (KAMMA.py:338-349, 397; every call
 * is commented out there, the runs are on the flat plane). With a terrain attached:
 *   - physics: each substep samples the field under the 4 wheel centres (the get_heights rule:
 *     min of the pixel and its +1,+1 neighbour) and the wheels stand on the plane through those
 *     4 footprints (docs/SPEC.md §3.8); OIS_BUF_PARAMS keeps mu, its plane normal is ignored;
 *   - pose buffers stay relative to the arena's env_origin (bounds check KAMMA.py:758-759);
 *   - curriculum != 0: update_SI_level (KAMMA.py:502-529) moves a resetting arena one level
 *     down / up when its finished episode's r1 sum is < threshold_low / > threshold_high and
 *     re-reads its origin from the table;
 *   - ois_height_scan = get_heights (KAMMA.py:872-893): 140 heights per arena around the base
 *     (grid KAMMA.py:278-292), the optional observation block of KAMMA.py:826-828.
 * The field and the table are caller-owned device memory and must outlive the attachment. */
#define OIS_HEIGHT_POINTS 140
typedef struct ois_terrain {
  uint32_t struct_bytes;     /* = sizeof(ois_terrain) */
  int32_t rows, cols;        /* Terrain.tot_rows, tot_cols (origin_terrain_generator.py:56-58) */
  float horizontal_scale;    /* 0.02  :39 */
  float vertical_scale;      /* 0.006 :40 */
  float border_size;         /* 20    :41; the field's pixel (0,0) sits at world (-border, -border) KAMMA.py:300 */
  int32_t num_levels, num_types; /* shape of the terrain_origins table (env_rows, env_cols :48-49) */
  int32_t curriculum;        /* terrain.curriculum KAMMA.py:265 */
  float threshold_low, threshold_high; /* 0.3, 5.0 KAMMA.py:509-510 */
  int32_t max_level;         /* 5 KAMMA.py:525 */
  int32_t footprint_stride;  /* 1 (0 reads as 1): the wheel footprints are sampled on the field at every substep. 2: at every
                                second substep; the substeps between keep the fitted plane (docs/SPEC.md section 3.8). Build-
                                defined (the reference's terrain is commented out, KAMMA.py:296-349): a cheaper contact, not
                                the same trajectories: a wheel meets an edge of the field up to one substep (<= 2 cm) later */
} ois_terrain;
/* heights_device: int16 [rows, cols]; terrain_origins_device: float [num_levels, num_types, 3];
 * levels_types_host: int32 [N, 2] initial terrain_levels / terrain_types (host memory, copied).
 * minfield_device: caller-owned scratch of ois_minfield_bytes(rows, cols) bytes the library fills (on
 *   `stream`) with m(x, y) = min(h[x][y], h[x+1][y+1])  — the two-sample rule of get_heights
 * (KAMMA.py:889-893) folded into the field once, so that a footprint or a scan point is ONE gather
 * instead of two (same values; the indices are clamped to rows - 2 / cols - 2 before use,
 * KAMMA.py:886-887, so the last row and column are never read) — stored in tiles of 8 x 8 cells
 * (one 128-byte line each; rows and cols rounded up to multiples of 8), because the points a robot
 * samples are a rotated 10 cm lattice: 2-3 of them share a tile, none a row-major line. Behind it, in
 * the same scratch, the field once more for ois_height_scan as 8-bit offsets from a per-tile minimum in
 * tiles of 16 x 8 cells (a line holds twice the cells; a tile spanning more than 254 height units is
 * marked and read from the 16-bit tiles: same values always). 256-byte aligned; the whole scratch must
 * stay below 2^31 bytes (OIS_EINVAL otherwise). Like the field it must outlive the attachment.
 * Call before ois_reset. terrain == NULL detaches (back to the per-arena plane). */
size_t ois_minfield_bytes(int rows, int cols);
int ois_terrain_attach(ois_env* env, const ois_terrain* terrain, const int16_t* heights_device,
                       int16_t* minfield_device, const float* terrain_origins_device,
                       const int32_t* levels_types_host, void* stream);
/* heights_out_device: float [N, OIS_HEIGHT_POINTS] measured_heights of the current state */
int ois_height_scan(ois_env* env, float* heights_out_device, void* stream);

/* Snapshot / restore the complete environment state (all persistent buffers) to/from a
 * host blob of ois_state_bytes(). Beyond reference parity (SURVEY.md §5 checkpoint).
 * The blob carries a hash of the ois_config it was saved under, the descriptor of the attached
 * terrain and the cursor of the attached transition memory; ois_load_state refuses (OIS_EINVAL, with
 * the reason in ois_last_error) a blob from another config, terrain attachment or memory size —
 * attach the same terrain / a memory of the same size first. The height field and the transition
 * tensors themselves are caller-owned and not part of the blob; the cursor (memory_index, filled)
 * is restored. */
size_t ois_state_bytes(const ois_env* env);
int ois_save_state(ois_env* env, void* host_blob, size_t bytes, void* stream);
int ois_load_state(ois_env* env, const void* host_blob, size_t bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* OIS_H_ */
