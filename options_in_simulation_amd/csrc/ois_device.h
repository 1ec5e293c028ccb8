// ois_device.h — device-side building blocks of the fused skid-steer step kernel (gfx950).
//
// Execution shape: ONE ARENA PER LANE (default). A lane owns the chassis and all four wheel
// contacts of its arena; the four contacts are independent inside a solver iteration
// (mass-splitting Jacobi, docs/SPEC.md §3.5). The kernel is bound by VALU issue, not by HBM
// (DESIGN.md §4). Measured on MI355X (tools/valu_rate.hip, profiles/r1_valu_rate.txt): a
// wave64 v_fma_f32 costs 2.0-2.4 ns per SIMD when the SIMD holds one wave and 1.1-1.3 ns with
// two or more, v_pk_fma_f32 2.5 / 1.9 ns, a quad_perm DPP add 1.75-2.0 ns. An earlier layout
// with one wheel per lane (4 lanes per arena, DPP quad sums) replicated the chassis math four
// times and paid 62 DPP adds per substep: 88 us per 65 536-arena step against 35-50 us here.
//
// Memory: every state array in HBM is [N][4] floats, so a lane reads its arena's row with one
// 16-byte global_load_dwordx4 and a wave-level instruction moves 1 KiB of contiguous bytes.
//
// Floating point: float32, operation order fixed by docs/SPEC.md; every fused multiply-add
// is an explicit __builtin_fmaf and the file is compiled with -ffp-contract=off
// -fno-slp-vectorize, so results are bit-identical to the CPU oracle's.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/ois.h"

namespace ois {

#define OIS_DEV __device__ __forceinline__

constexpr int kGroup = 16;  // arenas per reduction group of the reset-event records (SPEC §2.10)

// ---------------------------------------------------------------- kernel constants ----
// Three blocks of 32 / 32 / 48 dwords, by WHEN a policy step needs them. The step kernels are at the SGPR limit
// (106) with the physics constants alone: a by-value argument that is first read after the 20 substeps is loaded at
// kernel entry all the same, and then lives through the substep loop as v_writelane / v_readlane spill traffic
// (179 + ~130 VALU instructions per single-step launch, round 4's census). So the step kernels take only `phys`
// from the by-value argument and re-read `task` / `reset` from the kernarg segment where a phase starts
// (kernarg_late below); every other kernel uses the by-value fields as usual.
struct PhysK {   // the substeps (docs/SPEC.md §3): SGPR operands of the whole substep loop
  float dt, inv_dt, gdt, half_dt, im4, imq, J4x, J4y, J4z, dt_over_Is, dt_over_Iy, rw, rw_over_Is,
      rw2_over_Is, beta_inv_dt, vdep, maxw, maxw2, damping, wlim, climit, wx, wy, cz;
  float inv4wx, inv4wy;  // 1 / (4 wx), 1 / (4 wy): footprint-plane fit of the terrain contact (SPEC §3.8)
  int iters, nsub;
  int pad_[4];
};
struct TaskK {   // pre_physics_step / post_physics_step of a policy step (KAMMA.py constants, see include/ois.h)
  float Kp, Kp_omega, vehicle_mass, vehicle_inertia, policy_dt, action_scale, torque_clip, max_sim_steps,
      gamma1_const, gamma2_const, w1, w2, w3, sparse_bonus, sparse_threshold, terminal_reward,
      bounds, clip_obs, com_height;
  int action_dim, mix_mode, gamma1_mode, gamma2_mode, guiding_mode, cmd_sampler, eval_sums,
      max_episode_length, env0_reward_guard;
  uint32_t seed_lo, seed_hi;
  int64_t arena_offset;
};
struct ResetK {  // reset_idx and the per-level command generators: the paths a lane takes once per episode
  float episode_length_s, eval_ramp_steps;
  float cmd_x_lo, cmd_x_hi, cmd_y_lo, cmd_y_hi, cmd_yaw_lo, cmd_yaw_hi, cmd_x_span, cmd_y_span, cmd_yaw_span,
      spawn_xy_range;
  // scalars, no arrays: the step kernels hold this block as a LOCAL struct (kernarg_late), and one array member that the
  // optimiser decides to index (a select chain over its elements is enough) keeps the whole struct in memory — a private
  // array, which the backend promotes to LDS: +13 us per launch (tests/test_abi_and_config.py)
  float init_pos0, init_pos1, init_pos2, init_quat0, init_quat1, init_quat2, init_quat3;
  float init_lin_vel0, init_lin_vel1, init_lin_vel2, init_ang_vel0, init_ang_vel1, init_ang_vel2;
  float init_dof_vel0, init_dof_vel1, init_dof_vel2, init_dof_vel3;
  float cmd_thr_high;
  // per-level command generators (OIS_CMD_LEVEL): Python doubles of the reference kept as doubles
  double cmd_dt, cmd_episode_s;
  int pad_[14];
};
struct DevConsts {
  PhysK phys;
  TaskK task;
  ResetK reset;
};
static_assert(sizeof(PhysK) == 128 && sizeof(TaskK) == 128 && sizeof(ResetK) == 192, "blocks of 16-dword scalar loads");
static_assert(offsetof(DevConsts, task) == 128 && offsetof(DevConsts, reset) == 256 && sizeof(DevConsts) == 448, "kernarg offsets of kernarg_late");

// height-field terrain (include/ois.h ois_terrain_attach); on == 0: the per-arena plane of `params`
constexpr int kWideTile = -32768;  // DevTerrain::mb value of a tile whose heights do not fit base + 0..254
struct DevTerrain {
  const int16_t* h;        // [rows][cols] raw heights (caller-owned)
  const int16_t* m;        // [rows][cols] min(h[x][y], h[x+1][y+1]), filled at attach (caller-owned scratch)
  const uint8_t* m8;       // the same min-field as 8-bit offsets from a per-tile base, tiles of 16 x 8 cells = 128 B (height scan)
  const int16_t* mb;       // base of each 16 x 8 tile, in blocks of 8 x 8 tiles (base_index); kWideTile: range > 254, read m instead
  const float* origins;    // [levels][types][3] terrain_origins table (caller-owned)
  int rows, cols, ntypes, max_level, curriculum, on;
  int tiles_per_row;  // the min-field is stored in 8 x 8-cell tiles of 128 B: ceil(cols / 8) tiles per tile row
  float inv_hs, hs, vs, border, thr_lo, thr_hi;
};

// device pointers of one environment ([N][4] unless noted). Field ORDER matters: four blocks of 8 pointers
// (16 dwords each), by when a step kernel needs them, re-read from the kernarg segment at that point (kernarg_late).
struct RowPtrs {    // chassis rows a step stores in its epilogue
  float4 *pose, *quat, *linvel, *angvel, *commands, *sums_a, *sums_b;
  int4* counters;
};
struct WheelPtrs {  // wheel rows, eval sums, terrain rows (epilogue)
  float4 *wheel_vel, *wheel_torqs, *sums_c;
  uint2 *lam_n, *lam_l, *lam_t;  // [N] x 4 halves: warm-start impulses in binary16 (include/ois.h OIS_BUF_LAMBDA_*)
  float4* env_origins;           // terrain: per-arena tile origin (xyz)
  int4* terrain_idx;             // (level, type) index
};
struct OutPtrs {    // what post_physics_step leaves: outputs and the reset-event records
  float* obs;        // [N][6]
  float* rew;        // [N]
  uint8_t* done;     // [N]
  uint8_t* timeout;  // [N]
  float* env0_diag;  // [32]
  float* partials;   // [ngroups][32] per-group reset-event records
  int32_t* stamps;   // [ngroups]
  double* stat_acc;  // [ngroups][8] finished-episode accumulators, double
};
struct AuxPtrs {    // optional FIFO transition memory (include/ois.h ois_replay_*), level curriculum, logs
  float *rp_states, *rp_actions, *rp_rewards, *rp_next;  // [M][N][6|4|1|6]
  uint8_t* rp_done;                                      // [M][N]
  int32_t* reset_stamp;  // the sim_steps value of the last step in which any arena reset (two slots by parity)
  float* cmd_fraction;   // OIS_CMD_LEVEL: self.fraction [N]
  float* episode_info;   // [32]
};
struct DevBuffers {
  float4 *pose, *quat, *linvel, *angvel, *commands, *sums_a, *sums_b;
  int4* counters;
  float4 *wheel_vel, *wheel_torqs, *sums_c;
  uint2 *lam_n, *lam_l, *lam_t;
  float4* env_origins;
  int4* terrain_idx;
  float* obs;
  float* rew;
  uint8_t* done;
  uint8_t* timeout;
  float* env0_diag;
  float* partials;
  int32_t* stamps;
  double* stat_acc;
  float *rp_states, *rp_actions, *rp_rewards, *rp_next;
  uint8_t* rp_done;
  int32_t* reset_stamp;
  float* cmd_fraction;
  float* episode_info;
  float4* params;
  int rp_size;  // 0 when no transition memory is attached
  DevTerrain ter;
};
constexpr int kArgB = (int)sizeof(DevConsts);  // kernarg offset of the DevBuffers argument (second argument of every step shell)
static_assert(sizeof(RowPtrs) == 64 && sizeof(WheelPtrs) == 64 && sizeof(OutPtrs) == 64 && sizeof(AuxPtrs) == 64, "16-dword blocks");
static_assert(offsetof(DevBuffers, pose) == 0 && offsetof(DevBuffers, counters) == 56 && offsetof(DevBuffers, wheel_vel) == 64 &&
              offsetof(DevBuffers, terrain_idx) == 120 && offsetof(DevBuffers, obs) == 128 && offsetof(DevBuffers, stat_acc) == 184 &&
              offsetof(DevBuffers, rp_states) == 192 && offsetof(DevBuffers, episode_info) == 248 && alignof(DevBuffers) == 8 &&
              alignof(DevConsts) == 8, "DevBuffers = RowPtrs | WheelPtrs | OutPtrs | AuxPtrs | ...");

// A 16-dword block of the kernel's own arguments, read from the kernarg segment HERE (scalar load, waited for) instead of
// being held from kernel entry: volatile, so neither hoisted nor merged with the entry loads. T is one of the blocks above.
typedef int sgpr16_t __attribute__((ext_vector_type(16)));
typedef uint64_t kernarg_t;  // the kernarg segment pointer as an integer (an SGPR pair)
OIS_DEV kernarg_t kernarg_base() { return (kernarg_t)__builtin_amdgcn_kernarg_segment_ptr(); }
// ... and the pointer itself is not held through the substep loop either (the loop is two scalar registers short, and the
// allocator's choice of victim was a pair the loop READS: two reloads per substep): parked in two lanes of a vector register
// across the substeps, by hand
OIS_DEV int kernarg_park(kernarg_t p) {
  int v;
  asm volatile("v_writelane_b32 %0, %1, 0\n\tv_writelane_b32 %0, %2, 1" : "=&v"(v) : "s"((int)(uint32_t)p), "s"((int)(uint32_t)(p >> 32)));
  return v;
}
OIS_DEV kernarg_t kernarg_unpark(int v) {
  int lo, hi;
  asm volatile("v_readlane_b32 %0, %2, 0\n\tv_readlane_b32 %1, %2, 1\n\ts_nop 4" : "=&s"(lo), "=&s"(hi) : "v"(v));
  return (kernarg_t)(uint32_t)lo | ((kernarg_t)(uint32_t)hi << 32);
}
template <class T, int BYTE_OFFSET>
OIS_DEV T kernarg_late(kernarg_t kp) {
  static_assert(sizeof(T) == 64 && BYTE_OFFSET % 4 == 0, "one s_load_dwordx16");
  sgpr16_t v;
  asm volatile("s_load_dwordx16 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(kp), "n"(BYTE_OFFSET));
  return __builtin_bit_cast(T, v);
}
struct TaskHalves { sgpr16_t a, b; };
OIS_DEV TaskK late_task(kernarg_t kp) {
  static_assert(sizeof(TaskK) == 128, "");
  TaskHalves h;
  asm volatile("s_load_dwordx16 %0, %2, %3\n\ts_load_dwordx16 %1, %2, %4\n\ts_waitcnt lgkmcnt(0)"
               : "=&s"(h.a), "=&s"(h.b) : "s"(kp), "n"((int)offsetof(DevConsts, task)), "n"((int)offsetof(DevConsts, task) + 64));
  return __builtin_bit_cast(TaskK, h);
}
struct ResetThirds { sgpr16_t a, b, c; };
OIS_DEV ResetK late_reset(kernarg_t kp) {
  static_assert(sizeof(ResetK) == 192, "");
  ResetThirds h;
  asm volatile("s_load_dwordx16 %0, %3, %4\n\ts_load_dwordx16 %1, %3, %5\n\ts_load_dwordx16 %2, %3, %6\n\ts_waitcnt lgkmcnt(0)"
               : "=&s"(h.a), "=&s"(h.b), "=&s"(h.c)
               : "s"(kp), "n"((int)offsetof(DevConsts, reset)), "n"((int)offsetof(DevConsts, reset) + 64),
                 "n"((int)offsetof(DevConsts, reset) + 128));
  return __builtin_bit_cast(ResetK, h);
}

// ---------------------------------------------------------------- RNG -----------------
// Philox4x32-10 (Salmon et al., SC'11; Random123 constants)
OIS_DEV void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                           uint32_t k1, uint32_t o[4]) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    // one v_mad_u64_u32 per product (both halves) instead of a v_mul_hi_u32 + v_mul_lo_u32 pair
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    uint32_t h0 = (uint32_t)(p0 >> 32), l0 = (uint32_t)p0;
    uint32_t h1 = (uint32_t)(p1 >> 32), l1 = (uint32_t)p1;
    uint32_t n0 = h1 ^ c1 ^ k0, n2 = h0 ^ c3 ^ k1;
    c0 = n0; c1 = l1; c2 = n2; c3 = l0;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  o[0] = c0; o[1] = c1; o[2] = c2; o[3] = c3;
}
OIS_DEV float u01(uint32_t x) { return (float)(x >> 8) * 5.9604644775390625e-08f; }
enum { PURPOSE_MASK = 0, PURPOSE_RESET_A = 1, PURPOSE_RESET_B = 2, PURPOSE_ACTION = 3, PURPOSE_PARAMS = 4, PURPOSE_RESET_C = 5, PURPOSE_CMD = 6, PURPOSE_PUSH = 7 };
OIS_DEV void draw4(const TaskK& T, int64_t gid, uint32_t step, uint32_t purpose, float u[4]) {
  uint32_t o[4];
  philox4x32_10((uint32_t)((uint64_t)gid & 0xffffffffu), (uint32_t)((uint64_t)gid >> 32), step, purpose,
                T.seed_lo, T.seed_hi, o);
  u[0] = u01(o[0]); u[1] = u01(o[1]); u[2] = u01(o[2]); u[3] = u01(o[3]);
}

// deterministic sin/cos on [0, pi]: Cody-Waite by pi/2 + cephes minimax kernels (mul/add/fma only)
OIS_DEV void sincos_det(float x, float& s, float& c) {
  int k = (int)(x * 0.636619772367581343f + 0.5f);
  float kf = (float)k;
  float r = __builtin_fmaf(-kf, 1.57079637050628662109375f, x);
  r = __builtin_fmaf(-kf, -4.37113900018624283e-8f, r);
  float z = r * r;
  float ps = __builtin_fmaf(z, -1.9515295891e-4f, 8.3321608736e-3f);
  ps = __builtin_fmaf(z, ps, -1.6666654611e-1f);
  float sn = __builtin_fmaf(r * z, ps, r);
  float pc = __builtin_fmaf(z, 2.443315711809948e-5f, -1.388731625493765e-3f);
  pc = __builtin_fmaf(z, pc, 4.166664568298827e-2f);
  float cs = __builtin_fmaf(z * z, pc, __builtin_fmaf(-0.5f, z, 1.0f));
  int q = k & 3;
  s = q == 0 ? sn : (q == 1 ? cs : (q == 2 ? -sn : -cs));
  c = q == 0 ? cs : (q == 1 ? -sn : (q == 2 ? -cs : sn));
}

// Deterministic natural log on (0, 1] (cephes logf: frexp by bit manipulation + degree-8
// polynomial, mul/add/fma only) and a Box-Muller standard normal on top of it: the GridSampling
// command sampler (KAMMA_curr2.py:602-640, torch.normal there). Mirrors the oracle bit for bit.
OIS_DEV float log_det(float x) {
  uint32_t u = __float_as_uint(x);
  int e = (int)((u >> 23) & 0xffu) - 126;  // x = m * 2^e, m in [0.5, 1)
  float m = __uint_as_float((u & 0x007fffffu) | 0x3f000000u);
  if (m < 0.707106781186547524f) { e -= 1; m = (m + m) - 1.0f; } else { m = m - 1.0f; }
  const float z = m * m;
  float p = __builtin_fmaf(7.0376836292e-2f, m, -1.1514610310e-1f);
  p = __builtin_fmaf(p, m, 1.1676998740e-1f);
  p = __builtin_fmaf(p, m, -1.2420140846e-1f);
  p = __builtin_fmaf(p, m, 1.4249322787e-1f);
  p = __builtin_fmaf(p, m, -1.6668057665e-1f);
  p = __builtin_fmaf(p, m, 2.0000714765e-1f);
  p = __builtin_fmaf(p, m, -2.4999993993e-1f);
  p = __builtin_fmaf(p, m, 3.3333331174e-1f);
  float y = (p * m) * z;
  const float ef = (float)e;
  y = __builtin_fmaf(-2.12194440e-4f, ef, y);
  y = __builtin_fmaf(-0.5f, z, y);
  return __builtin_fmaf(0.693359375f, ef, m + y);
}
OIS_DEV float normal_det(float u1, float u2) {  // u1, u2 in [0, 1) -> N(0, 1)
  const float r = sqrtf(-2.0f * log_det(1.0f - u1));
  const float t = u2 < 0.5f ? u2 : 1.0f - u2;  // cos(2 pi t) is even about t = 0.5
  float s, c;
  sincos_det(6.28318530717958647692f * t, s, c);
  return r * c;
}

// Deterministic sin for |x| < ~1e4 (cephes sinf reduction by pi/4, the kernels of sincos_det) and the
// per-terrain-level command generators of sample_velocity_command (KAMMA.py:533-567): mirrors of
// the oracle's orc_sin / orc_level_command, bit for bit (doubles where the reference has Python doubles).
OIS_DEV float sin_det(float x) {
  float sign = 1.0f;
  if (x < 0.0f) { x = -x; sign = -1.0f; }
  int j = (int)(1.27323954473516f * x);
  float y = (float)j;
  if (j & 1) { j += 1; y += 1.0f; }
  j &= 7;
  if (j > 3) { sign = -sign; j -= 4; }
  const float r = __builtin_fmaf(-y, 3.77489497744594108e-8f, __builtin_fmaf(-y, 2.4187564849853515625e-4f, __builtin_fmaf(-y, 0.78515625f, x)));
  const float z = r * r;
  float v;
  if (j == 1 || j == 2) {
    float pc = __builtin_fmaf(z, 2.443315711809948e-5f, -1.388731625493765e-3f);
    pc = __builtin_fmaf(z, pc, 4.166664568298827e-2f);
    v = __builtin_fmaf(z * z, pc, __builtin_fmaf(-0.5f, z, 1.0f));
  } else {
    float ps = __builtin_fmaf(z, -1.9515295891e-4f, 8.3321608736e-3f);
    ps = __builtin_fmaf(z, ps, -1.6666654611e-1f);
    v = __builtin_fmaf(r * z, ps, r);
  }
  return sign * v;
}
OIS_DEV void level_command(int level, float fraction, int step_counter, float z, double dt, double episode_length_s, float& cmd_x,
                           float& cmd_yaw) {
  if (level == 0) {
    const float sigma = 0.01f + 0.09f * fraction;  // :538
    const float x = 0.5f + sigma * z;               // :539
    cmd_x = 0.0f; cmd_yaw = x > 0.0f ? x : 0.0f;    // :541 return 0.0, max(x_vel, 0.0)
  } else if (level == 1) {
    const float freq = 0.01f + 0.09f * fraction;   // :546
    const double t = (double)step_counter * dt;     // :550
    const float sn = sin_det(freq * (float)t);      // :551
    float x;
    if (fraction > 0.5f) { const float amp = 0.1f + 0.4f * fraction; x = 0.5f + amp * sn; }  // :548-549
    else x = (float)(0.5 + 0.1 * (double)sn);
    cmd_x = x > 0.0f ? x : 0.0f; cmd_yaw = 0.0f;
  } else if (level == 2) {
    const double t = (double)step_counter * dt;     // :558
    const float nz = 0.0f + (0.5f * fraction) * z;  // :560-561 torch.normal(mean=0.0, std=Noise)
    const float x = (float)((double)nz + 0.5 * t / episode_length_s);
    cmd_x = x > 0.0f ? x : 0.0f; cmd_yaw = 0.0f;
  } else {
    const float x = 0.5f + 0.1f * z;                // :566
    cmd_x = x > 0.0f ? x : 0.0f; cmd_yaw = 0.0f;
  }
}

// ---------------------------------------------------------------- quaternions ---------
// omni.isaac.core.utils.torch.rotations quat_apply / quat_mul (wxyz), call sites
// KAMMA.py:434, 701, 749
OIS_DEV void quat_apply(const float q[4], float vx, float vy, float vz, float o[3]) {
  float w = q[0], x = q[1], y = q[2], z = q[3];
  float tx = (y * vz - z * vy) * 2.0f;
  float ty = (z * vx - x * vz) * 2.0f;
  float tz = (x * vy - y * vx) * 2.0f;
  o[0] = (vx + w * tx) + (y * tz - z * ty);
  o[1] = (vy + w * ty) + (z * tx - x * tz);
  o[2] = (vz + w * tz) + (x * ty - y * tx);
}
OIS_DEV void quat_mul(const float a[4], const float b[4], float o[4]) {
  float w1 = a[0], x1 = a[1], y1 = a[2], z1 = a[3];
  float w2 = b[0], x2 = b[1], y2 = b[2], z2 = b[3];
  float ww = (z1 + x1) * (x2 + y2);
  float yy = (w1 - y1) * (w2 + z2);
  float zz = (w1 + y1) * (w2 - z2);
  float xx = ww + yy + zz;
  float qq = 0.5f * (xx + (z1 - x1) * (x2 - y2));
  o[0] = qq - ww + (z1 - y1) * (y2 - z2);
  o[1] = qq - xx + (x1 + w1) * (x2 + w2);
  o[2] = qq - yy + (w1 - x1) * (y2 + z2);
  o[3] = qq - zz + (z1 + y1) * (w2 - x2);
}
OIS_DEV void rotmat(const float q[4], float R[9]) {
  float qw = q[0], qx = q[1], qy = q[2], qz = q[3];
  float x2 = qx + qx, y2 = qy + qy, z2 = qz + qz;
  float xx = qx * x2, yy = qy * y2, zz = qz * z2;
  float xy = qx * y2, xz = qx * z2, yz = qy * z2;
  float wx = qw * x2, wy = qw * y2, wz = qw * z2;
  R[0] = 1.0f - (yy + zz); R[1] = xy - wz;          R[2] = xz + wy;
  R[3] = xy + wz;          R[4] = 1.0f - (xx + zz); R[5] = yz - wx;
  R[6] = xz - wy;          R[7] = yz + wx;          R[8] = 1.0f - (xx + yy);
}
OIS_DEV void rotT(const float R[9], const float a[3], float o[3]) {  // o = R^T a
  o[0] = __builtin_fmaf(R[6], a[2], __builtin_fmaf(R[3], a[1], R[0] * a[0]));
  o[1] = __builtin_fmaf(R[7], a[2], __builtin_fmaf(R[4], a[1], R[1] * a[0]));
  o[2] = __builtin_fmaf(R[8], a[2], __builtin_fmaf(R[5], a[1], R[2] * a[0]));
}
OIS_DEV void rot(const float R[9], const float a[3], float o[3]) {  // o = R a
  o[0] = __builtin_fmaf(R[2], a[2], __builtin_fmaf(R[1], a[1], R[0] * a[0]));
  o[1] = __builtin_fmaf(R[5], a[2], __builtin_fmaf(R[4], a[1], R[3] * a[0]));
  o[2] = __builtin_fmaf(R[8], a[2], __builtin_fmaf(R[7], a[1], R[6] * a[0]));
}

OIS_DEV float clipf(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }

// binary16 storage of the warm-start impulses: v_cvt_f16_f32 rounds to nearest even (default mode)
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
OIS_DEV float round_half(float x) { return (float)(_Float16)x; }
OIS_DEV uint32_t pack_half2(float lo, float hi) {
  half2_t v;
  v.x = (_Float16)lo; v.y = (_Float16)hi;
  return __builtin_bit_cast(uint32_t, v);
}
OIS_DEV void unpack_half2(uint32_t u, float& lo, float& hi) {
  const half2_t v = __builtin_bit_cast(half2_t, u);
  lo = (float)v.x; hi = (float)v.y;
}

// ---------------------------------------------------------------- lanes per arena -----
// LPA = 1: one lane owns the whole arena (4 wheels).  LPA = 2: two adjacent lanes share an
// arena, lane h = tid & 1 owns the wheels {2h, 2h+1} = the left (fl, rl) or right (fr, rr)
// side; chassis state is replicated in both lanes. At 65 536 arenas LPA = 2 puts two waves on
// every SIMD (a single wave per SIMD issues a VALU instruction only every ~1.5-1.9 ns against
// ~1.2 ns with two, profiles/r1_valu_rate.txt) at the price of computing the chassis part
// twice and of one quad_perm DPP add per cross-side sum. Wheel sums keep the SPEC order
// (w0 + w1) + (w2 + w3): each lane adds its own pair, then the partner's pair.
template <int CTRL>
OIS_DEV float dpp(float x) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xF, 0xF, true));
}
template <int LPA>
OIS_DEV float side_sum(float own) {  // own = sum over this lane's wheels
  if (LPA == 1) return own;
  return own + dpp<0xB1>(own);  // quad_perm [1,0,3,2]: partner lane; a + b == b + a bitwise
}
template <int LPA>
OIS_DEV float sum_wheels(const float* x) {  // x[4 / LPA]
  if (LPA == 1) return (x[0] + x[1]) + (x[2] + x[3]);
  return side_sum<2>(x[0] + x[1]);
}

// ---------------------------------------------------------------- per-arena state -----
template <int LPA>
struct ArenaT {
  static constexpr int WPL = 4 / LPA;  // wheels per lane
  float p[3], vfwd;      // COM position, v_forward_projected of the previous step
  float q[4];
  float v[3], prev_lin;  // world linear velocity, previous_linear_velocity[:,0]
  float w[3], prev_ang;  // world angular velocity, previous_angular_velocity[:,2]
  float cmd[4];
  int ep_buf, sim_steps, ep_count;
  float mu, n[3];
  float jv[WPL];                      // wheel joint velocities (dof_vel) of this lane's wheels
  float ln[WPL], l1[WPL], l2[WPL];    // cached contact impulses
  float wt[WPL];                      // clipped wheel torques (wheel_torqs)
  float sa[4], sb[4], sc[4];          // episode sums: r1 r2 r3 dense | sparse guiding observed final | eval
};

// torch_rand_float(lower, upper) = (upper - lower) * rand + lower  (KAMMA.py:424-456)
OIS_DEV float rand_float(float lo, float hi, float u) { return (hi - lo) * u + lo; }
// same with the span (upper - lower) precomputed in double by the host, as Python does
OIS_DEV float rand_span(float lo, float span, float u) { return span * u + lo; }

// a11: KAMMA.py:701-710
OIS_DEV float forward_projection(const float q[4], const float v[3]) {
  float f[3];
  quat_apply(q, 1.0f, 0.0f, 0.0f, f);
  float nrm = sqrtf((f[0] * f[0] + f[1] * f[1]) + f[2] * f[2]);
  float fx = f[0] / nrm, fy = f[1] / nrm, fz = f[2] / nrm;
  return (v[0] * fx + v[1] * fy) + v[2] * fz;
}

// What a resetting arena contributes to its group's reset-event record (SPEC §2.10).
// Record slots: 0 count, 1 return^2, 2 length, 4..7 sums r1 r2 r3 dense, 8..11 sums sparse
// guiding observed final, 12..15 ratios r1 r2 r3 dense, 16..18 ratios sparse guiding observed,
// 20..22 eval sums, 24 gamma1, 25 gamma2, 26 -return (max), 27 return (max)
constexpr int kRecSum = 23;  // slots [0, 23) are summed; 24..27 are max-reduced
struct ResetContrib {
  float s[kRecSum];
  float m[4];
};
OIS_DEV void contrib_identity(ResetContrib& c) {
#pragma unroll
  for (int k = 0; k < kRecSum; ++k) c.s[k] = 0.0f;
#pragma unroll
  for (int k = 0; k < 4; ++k) c.m[k] = -__builtin_inff();
}

// reset_idx for one arena (KAMMA.py:417-498). wheel0 = index of this lane's first wheel. ALIGN: S.n holds the arena's plane (below).
template <int LPA, bool ALIGN>
OIS_DEV void reset_arena(const TaskK& T, const ResetK& Rk, int64_t gid, int wheel0, float gamma1, float gamma2, ArenaT<LPA>& S,
                         ResetContrib& rc) {
  // by value, read at constant offsets HERE: behind the reference the optimiser folds the select chain below into one
  // indexed load, which keeps the caller's local copy of the block in memory (see ResetK)
  const float dv0 = Rk.init_dof_vel0, dv1 = Rk.init_dof_vel1, dv2 = Rk.init_dof_vel2, dv3 = Rk.init_dof_vel3;
  const float Sf = S.sb[3];                   // "Final reward" sum
  const float hg = 100.0f * (1.0f - gamma2);  // :470
  const float gg = 100.0f * gamma2;           // :475
  const float L = Rk.episode_length_s;
  rc.s[0] = 1.0f; rc.s[1] = Sf * Sf; rc.s[2] = (float)S.ep_buf; rc.s[3] = 0.0f;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    rc.s[4 + k] = S.sa[k];
    rc.s[8 + k] = S.sb[k];
    rc.s[12 + k] = L * ((hg * S.sa[k]) / Sf);  // :470-473
  }
  rc.s[16] = L * ((hg * S.sb[0]) / Sf);  // :474
  rc.s[17] = L * ((gg * S.sb[1]) / Sf);  // :475 guiding uses gamma2 itself
  rc.s[18] = L * ((hg * S.sb[2]) / Sf);  // :476
  rc.s[19] = 0.0f;
  rc.s[20] = S.sc[0]; rc.s[21] = S.sc[1]; rc.s[22] = S.sc[2];
  rc.m[0] = gamma1; rc.m[1] = gamma2; rc.m[2] = -Sf; rc.m[3] = Sf;
#pragma unroll
  for (int k = 0; k < 4; ++k) { S.sa[k] = 0.0f; S.sb[k] = 0.0f; S.sc[k] = 0.0f; }  // :485 (sc[3]: old-torque flag)

  float u[8];
  draw4(T, gid, (uint32_t)S.sim_steps, PURPOSE_RESET_A, &u[0]);
  draw4(T, gid, (uint32_t)S.sim_steps, PURPOSE_RESET_B, &u[4]);
  const float r = Rk.spawn_xy_range;
  const float px = Rk.init_pos0 + rand_float(-r, r, u[0]);  // :422-424
  const float py = Rk.init_pos1 + rand_float(-r, r, u[1]);
  const float pz = Rk.init_pos2;
  const float ang = rand_float(0.0f, 6.28318530717958647692f, u[2]);  // :426
  float s, c;
  sincos_det(ang / 2.0f, s, c);
  const float qz[4] = {c, 0.0f, 0.0f, s};
  const float iq[4] = {Rk.init_quat0, Rk.init_quat1, Rk.init_quat2, Rk.init_quat3};
  quat_mul(qz, iq, S.q);  // :434
  const float cx = rand_span(Rk.cmd_x_lo, Rk.cmd_x_span, u[3]);      // :448
  const float cy = rand_span(Rk.cmd_y_lo, Rk.cmd_y_span, u[4]);      // :451
  const float ch = rand_span(Rk.cmd_yaw_lo, Rk.cmd_yaw_span, u[5]);  // :454
  const float keep = sqrtf(cx * cx + cy * cy) > 0.25f ? 1.0f : 0.0f;  // :458
  S.cmd[1] = cy * keep;
  S.cmd[3] = ch * keep;
  const float cmd2_kept = S.cmd[2] * keep;  // :458 on the value commands[:, 2] had (only the level mode keeps it)
  float x_vel = rand_span(Rk.cmd_x_lo, Rk.cmd_x_span, u[6]);  // :571 (u[7]: discarded omega draw :572)
  if (T.cmd_sampler == OIS_CMD_BOX) {  // KAMMA_curr1.py:577-598
    const float progress = fminf(1.0f, (float)S.sim_steps / T.max_sim_steps);
    const float factor = 0.1f + (1.0f - 0.1f) * progress;
    const float xc = (Rk.cmd_x_lo + Rk.cmd_x_hi) / 2.0f;
    const float xmin = xc - (xc - Rk.cmd_x_lo) * factor;
    const float xmax = xc + (Rk.cmd_x_hi - xc) * factor;
    x_vel = rand_float(xmin, xmax, u[6]);
  } else if (T.cmd_sampler == OIS_CMD_EVAL) {  // KAMMA_eval.py:545-550
    const float progress = fminf(fmaxf((float)S.sim_steps / Rk.eval_ramp_steps, 0.0f), 1.0f);
    x_vel = 1.0f + progress * (2.0f - 1.0f);
  } else if (T.cmd_sampler == OIS_CMD_GRID) {  // KAMMA_curr2.py:602-640 (bimodal normal, sigma 0.05)
    const float progress = fminf(1.0f, (float)S.sim_steps / T.max_sim_steps);
    const float xc = (Rk.cmd_x_lo + Rk.cmd_x_hi) / 2.0f;
    const float left = xc - progress * (xc - Rk.cmd_x_lo);
    const float right = xc + progress * (Rk.cmd_x_hi - xc);
    float uc[4];
    draw4(T, gid, (uint32_t)S.sim_steps, PURPOSE_RESET_C, uc);
    const float mean = u[6] < 0.5f ? left : right;  // torch.rand(1) < 0.5 picks the side
    x_vel = mean + 0.05f * normal_det(uc[0], uc[1]);
  }
  S.cmd[0] = x_vel > 0.0f ? x_vel : 0.0f;  // :575, :495
  S.cmd[2] = 0.0f;                         // :574, :496
  if (T.cmd_sampler == OIS_CMD_LEVEL) {    // `if not self.curriculum:` skips the re-draw (:490)
    S.cmd[0] = cx * keep;
    S.cmd[2] = cmd2_kept;
  }
  // ALIGN (the arena's own plane of OIS_BUF_PARAMS; not the FLAT shells, whose plane is z = 0, nor a height field): the robot is set
  // down ON the plane — base link at its height above the ground point under (x, y), along n; the drawn orientation turned by the
  // shortest rotation from world z to n (plain Hamilton product: exact for n = (0, 0, 1)). Oracle reset_arena; SPEC 3.2.
  float bx = px, by = py, bz = pz;
  if constexpr (ALIGN) {
    const float nx = S.n[0], ny = S.n[1], nz = S.n[2], h = pz;
    const float zg = -(__builtin_fmaf(ny, py, nx * px) / nz);
    const float a = 1.0f + nz, d = sqrtf(a + a);
    const float w1 = a / d, x1 = -ny / d, y1 = nx / d;
    const float w2 = S.q[0], x2 = S.q[1], y2 = S.q[2], z2 = S.q[3];
    S.q[0] = __builtin_fmaf(-y1, y2, __builtin_fmaf(-x1, x2, w1 * w2));
    S.q[1] = __builtin_fmaf(y1, z2, __builtin_fmaf(x1, w2, w1 * x2));
    S.q[2] = __builtin_fmaf(-x1, z2, __builtin_fmaf(y1, w2, w1 * y2));
    S.q[3] = __builtin_fmaf(x1, y2, __builtin_fmaf(-y1, x2, w1 * z2));
    bx = __builtin_fmaf(h, nx, px);
    by = __builtin_fmaf(h, ny, py);
    bz = __builtin_fmaf(h, nz, zg);
  }
  float R[9];
  rotmat(S.q, R);
  S.p[0] = __builtin_fmaf(R[2], T.com_height, bx);
  S.p[1] = __builtin_fmaf(R[5], T.com_height, by);
  S.p[2] = __builtin_fmaf(R[8], T.com_height, bz);
  S.v[0] = Rk.init_lin_vel0; S.v[1] = Rk.init_lin_vel1; S.v[2] = Rk.init_lin_vel2;  // :435
  S.w[0] = Rk.init_ang_vel0; S.w[1] = Rk.init_ang_vel1; S.w[2] = Rk.init_ang_vel2;
#pragma unroll
  for (int k = 0; k < ArenaT<LPA>::WPL; ++k) {
    const int gk = wheel0 + k;  // :438
    S.jv[k] = gk == 0 ? dv0 : (gk == 1 ? dv1 : (gk == 2 ? dv2 : dv3));
    S.ln[k] = 0.0f; S.l1[k] = 0.0f; S.l2[k] = 0.0f;
  }
  S.ep_buf = 0;     // :462
  S.ep_count += 1;  // :463
}

// Pairwise butterfly over the 16 arenas (= 16 * LPA consecutive lanes) of a reduction group;
// with LPA = 2 both lanes of an arena hold the same value, so the butterfly skips lane bit 0.
template <int OP, int LPA>
OIS_DEV float group_reduce16(float x) {
#pragma unroll
  for (int m = LPA; m < kGroup * LPA; m <<= 1) {
    const float y = __shfl_xor(x, m, 64);
    x = OP == 0 ? x + y : fmaxf(x, y);
  }
  return x;
}

// Record of a reset event + finished-episode accumulators for the group of this lane.
// Must be called by all 64 lanes of the wave (shuffles); only groups with a resetting arena
// write. group = arena / 16.
template <int LPA>
OIS_DEV void group_record(float* partials, int32_t* stamps, double* stat_acc, int group, int lane, bool group_has_reset,
                          const ResetContrib& rc, int stamp, bool count_stats) {
  float rs[kRecSum], rm[4];
#pragma unroll
  for (int k = 0; k < kRecSum; ++k) rs[k] = group_reduce16<0, LPA>(rc.s[k]);
#pragma unroll
  for (int k = 0; k < 4; ++k) rm[k] = group_reduce16<1, LPA>(rc.m[k]);
  if ((lane & (kGroup * LPA - 1)) == 0 && group_has_reset) {
    float* P = partials + (size_t)group * 32;
#pragma unroll
    for (int k = 0; k < kRecSum; ++k) P[k] = rs[k];
#pragma unroll
    for (int k = 0; k < 4; ++k) P[24 + k] = rm[k];
    stamps[group] = stamp;
    if (count_stats) {
      double* A = stat_acc + (size_t)group * 8;
      // A[0..3] = count, ret^2, len, ret ; A[6], A[7] = max(-ret), max(ret)
      A[0] += (double)rs[0]; A[1] += (double)rs[1]; A[2] += (double)rs[2]; A[3] += (double)rs[11];
      A[6] = fmax(A[6], (double)rm[2]); A[7] = fmax(A[7], (double)rm[3]);
    }
  }
}

// does the 16-arena group of this lane contain a set bit of the wave ballot?
template <int LPA>
OIS_DEV bool group_any(unsigned long long ballot, int lane) {
  constexpr int span = kGroup * LPA;  // 16 or 32 lanes
  const unsigned long long mask = span == 64 ? ~0ull : ((1ull << span) - 1ull);
  return ((ballot >> (lane & ~(span - 1))) & mask) != 0ull;
}

}  // namespace ois
