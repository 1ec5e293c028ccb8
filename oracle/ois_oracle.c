/*
 * ois_oracle.c — CPU ORACLE (test infrastructure, NOT product code). See ois_oracle.h for
 * the parity status. Scalar float32, one arena at a time, operation order fixed by
 * docs/SPEC.md: every fused multiply-add is an explicit fmaf() and the file is compiled
 * with -ffp-contract=off, so the HIP kernel (which spells the same fmaf()s) can be
 * compared bit for bit.
 *
 * Reference citations are to /root/reference/src/my_envs/KAMMA.py unless another file is
 * named.
 */
#include "ois_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

static __thread char g_err[256];
const char* orc_last_error(void) { return g_err; }
static int fail(int code, const char* msg) {
  snprintf(g_err, sizeof g_err, "%s", msg);
  return code;
}

/* ------------------------------------------------------------------ RNG -------------- */
/* Philox4x32-10, Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3"
 * (SC'11); constants as in Random123 philox.h. */
void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
  uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
  uint32_t k0 = key[0], k1 = key[1];
  for (int r = 0; r < 10; ++r) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    uint32_t n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/* 24 random bits -> [0,1), exact in float32 */
float orc_u01(uint32_t x) { return (float)(x >> 8) * 5.9604644775390625e-08f; }

enum { PURPOSE_MASK = 0, PURPOSE_RESET_A = 1, PURPOSE_RESET_B = 2, PURPOSE_ACTION = 3, PURPOSE_PARAMS = 4, PURPOSE_RESET_C = 5, PURPOSE_CMD = 6, PURPOSE_PUSH = 7 };

static void draw4(uint64_t seed, int64_t arena, uint32_t step, uint32_t purpose, float u[4]) {
  uint32_t ctr[4] = {(uint32_t)((uint64_t)arena & 0xffffffffu), (uint32_t)((uint64_t)arena >> 32), step, purpose};
  uint32_t key[2] = {(uint32_t)(seed & 0xffffffffu), (uint32_t)(seed >> 32)};
  uint32_t o[4];
  orc_philox4x32_10(ctr, key, o);
  for (int i = 0; i < 4; ++i) u[i] = orc_u01(o[i]);
}

/* ------------------------------------------------------------------ binary16 --------- */
/* IEEE binary16 <-> binary32, round to nearest even, subnormals kept (what v_cvt_f16_f32 /
 * v_cvt_f32_f16 do in the kernel's default mode): the contact impulses carried from one policy
 * step to the next are stored in half precision (OIS_BUF_LAMBDA_*). */
uint16_t orc_f2h(float f) {
  uint32_t x;
  memcpy(&x, &f, 4);
  uint16_t sign = (uint16_t)((x >> 16) & 0x8000u);
  x &= 0x7fffffffu;
  if (x > 0x7f800000u) return (uint16_t)(sign | 0x7e00u);  /* NaN */
  if (x >= 0x47800000u) return (uint16_t)(sign | 0x7c00u); /* >= 2^16 (and inf) */
  if (x < 0x38800000u) {                                    /* below 2^-14: subnormal half or zero */
    if (x < 0x33000000u) return sign;                       /* below 2^-25: rounds to zero */
    uint32_t e = x >> 23, m = (x & 0x7fffffu) | 0x800000u;
    uint32_t shift = 126u - e;                              /* 14 .. 24 */
    uint32_t h = m >> shift, rem = m & ((1u << shift) - 1u), half = 1u << (shift - 1u);
    if (rem > half || (rem == half && (h & 1u))) h++;
    return (uint16_t)(sign | h);
  }
  uint32_t h = (x - 0x38000000u) >> 13, rem = x & 0x1fffu;
  if (rem > 0x1000u || (rem == 0x1000u && (h & 1u))) h++;  /* a carry into the exponent is the right answer, 65520 -> inf too */
  return (uint16_t)(sign | h);
}
float orc_h2f(uint16_t h) {
  uint32_t sign = (uint32_t)(h & 0x8000u) << 16, e = (h >> 10) & 0x1fu, m = h & 0x3ffu, x;
  if (e == 0) {
    if (m == 0) x = sign;
    else { /* subnormal: normalise */
      int s = 0;
      while (!(m & 0x400u)) { m <<= 1; s++; }
      x = sign | ((uint32_t)(113 - s) << 23) | ((m & 0x3ffu) << 13);
    }
  } else if (e == 31) x = sign | 0x7f800000u | (m << 13);
  else x = sign | ((e + 112u) << 23) | (m << 13);
  float f;
  memcpy(&f, &x, 4);
  return f;
}
static inline float round_half(float x) { return orc_h2f(orc_f2h(x)); }

/* ------------------------------------------------------------------ trig ------------- */
/* Deterministic sin/cos for x in [0, pi] built from mul/add/fma only (so the GPU can repeat
 * it bit for bit): Cody-Waite reduction by pi/2, cephes single-precision minimax kernels. */
void orc_sincos(float x, float* s, float* c) {
  int k = (int)(x * 0.636619772367581343f + 0.5f);
  float kf = (float)k;
  float r = fmaf(-kf, 1.57079637050628662109375f, x);
  r = fmaf(-kf, -4.37113900018624283e-8f, r);
  float z = r * r;
  float ps = fmaf(z, -1.9515295891e-4f, 8.3321608736e-3f);
  ps = fmaf(z, ps, -1.6666654611e-1f);
  float sn = fmaf(r * z, ps, r);
  float pc = fmaf(z, 2.443315711809948e-5f, -1.388731625493765e-3f);
  pc = fmaf(z, pc, 4.166664568298827e-2f);
  float cs = fmaf(z * z, pc, fmaf(-0.5f, z, 1.0f));
  switch (k & 3) {
    case 0: *s = sn; *c = cs; break;
    case 1: *s = cs; *c = -sn; break;
    case 2: *s = -sn; *c = -cs; break;
    default: *s = -cs; *c = sn; break;
  }
}

/* Deterministic sin for any |x| < ~1e4 (cephes sinf: reduction by pi/4 with a three-part pi/4,
 * the minimax kernels of orc_sincos; mul/add/fma and one float -> int truncation only). Used by the
 * level-1 command generator (KAMMA.py:550-551: math.sin(freq * t), t up to max_sim_steps * dt). */
float orc_sin(float x) {
  float sign = 1.0f;
  if (x < 0.0f) { x = -x; sign = -1.0f; }
  int j = (int)(1.27323954473516f * x); /* integer part of x / (pi/4) */
  float y = (float)j;
  if (j & 1) { j += 1; y += 1.0f; }     /* map zeros to the origin */
  j &= 7;
  if (j > 3) { sign = -sign; j -= 4; }
  float r = fmaf(-y, 3.77489497744594108e-8f, fmaf(-y, 2.4187564849853515625e-4f, fmaf(-y, 0.78515625f, x)));
  float z = r * r, v;
  if (j == 1 || j == 2) {
    float pc = fmaf(z, 2.443315711809948e-5f, -1.388731625493765e-3f);
    pc = fmaf(z, pc, 4.166664568298827e-2f);
    v = fmaf(z * z, pc, fmaf(-0.5f, z, 1.0f));
  } else {
    float ps = fmaf(z, -1.9515295891e-4f, 8.3321608736e-3f);
    ps = fmaf(z, ps, -1.6666654611e-1f);
    v = fmaf(r * z, ps, r);
  }
  return sign * v;
}

/* Deterministic natural log for x in (0, 1] (cephes logf: frexp by bit manipulation + degree-8
 * minimax polynomial, mul/add/fma only) and a Box-Muller standard normal built on it. Used by
 * the GridSampling command sampler only (KAMMA_curr2.py:602-640, torch.normal there). */
static float log_det(float x) {
  union { float f; uint32_t u; } v = {x};
  int e = (int)((v.u >> 23) & 0xffu) - 126; /* x = m * 2^e, m in [0.5, 1) */
  v.u = (v.u & 0x007fffffu) | 0x3f000000u;
  float m = v.f;
  if (m < 0.707106781186547524f) { e -= 1; m = (m + m) - 1.0f; } else { m = m - 1.0f; }
  float z = m * m;
  float p = fmaf(7.0376836292e-2f, m, -1.1514610310e-1f);
  p = fmaf(p, m, 1.1676998740e-1f);
  p = fmaf(p, m, -1.2420140846e-1f);
  p = fmaf(p, m, 1.4249322787e-1f);
  p = fmaf(p, m, -1.6668057665e-1f);
  p = fmaf(p, m, 2.0000714765e-1f);
  p = fmaf(p, m, -2.4999993993e-1f);
  p = fmaf(p, m, 3.3333331174e-1f);
  float y = (p * m) * z;
  float ef = (float)e;
  y = fmaf(-2.12194440e-4f, ef, y);
  y = fmaf(-0.5f, z, y);
  return fmaf(0.693359375f, ef, m + y);
}
/* u1, u2 uniform in [0, 1) -> N(0, 1): sqrt(-2 ln(1 - u1)) * cos(2 pi u2) */
static float normal_det(float u1, float u2) {
  float r = sqrtf(-2.0f * log_det(1.0f - u1));
  float t = u2 < 0.5f ? u2 : 1.0f - u2; /* cos(2 pi t) is even about t = 0.5 */
  float s, c;
  orc_sincos(6.28318530717958647692f * t, &s, &c);
  return r * c;
}

/* ------------------------------------------------------------------ quaternions ------ */
/* omni.isaac.core.utils.torch.rotations (third-party, not in the reference tree; formulas
 * restated from the published isaacgym torch_utils of the same name), wxyz convention.
 * call sites KAMMA.py:434 (quat_mul), :701, :707, :749 (quat_apply). */
void orc_quat_apply(const float q[4], const float v[3], float out[3]) {
  /* t = 2 * cross(xyz, v); out = v + w * t + cross(xyz, t) */
  float x = q[1], y = q[2], z = q[3], w = q[0];
  float tx = (y * v[2] - z * v[1]) * 2.0f;
  float ty = (z * v[0] - x * v[2]) * 2.0f;
  float tz = (x * v[1] - y * v[0]) * 2.0f;
  out[0] = (v[0] + w * tx) + (y * tz - z * ty);
  out[1] = (v[1] + w * ty) + (z * tx - x * tz);
  out[2] = (v[2] + w * tz) + (x * ty - y * tx);
}

void orc_quat_mul(const float a[4], const float b[4], float out[4]) {
  float w1 = a[0], x1 = a[1], y1 = a[2], z1 = a[3];
  float w2 = b[0], x2 = b[1], y2 = b[2], z2 = b[3];
  float ww = (z1 + x1) * (x2 + y2);
  float yy = (w1 - y1) * (w2 + z2);
  float zz = (w1 + y1) * (w2 - z2);
  float xx = ww + yy + zz;
  float qq = 0.5f * (xx + (z1 - x1) * (x2 - y2));
  out[0] = qq - ww + (z1 - y1) * (y2 - z2);
  out[1] = qq - xx + (x1 + w1) * (x2 + w2);
  out[2] = qq - yy + (w1 - x1) * (y2 + z2);
  out[3] = qq - zz + (z1 + y1) * (w2 - x2);
}

/* ------------------------------------------------------------------ a2 .. a11 -------- */
/* a2: KAMMA.py:604-624. dt is the POLICY dt 0.1 (self.dt, KAMMA.py:260). */
static inline void criteria1(float cmd_x, float cmd_yaw, float v_fwd_prev, float omega_z, float Kp,
                             float Kp_omega, float mass, float inertia, float dt, float* v_delta,
                             float* omega_delta, float* ac_left, float* ac_right) {
  float vd = cmd_x - v_fwd_prev;  /* :609 */
  float od = cmd_yaw - omega_z;   /* :610 */
  float lin = Kp * (mass * (vd / dt));            /* :617 first term  */
  float ang = Kp_omega * (inertia * (od / dt));   /* :617 second term */
  *v_delta = vd;
  *omega_delta = od;
  *ac_left = lin - ang;   /* :617 */
  *ac_right = lin + ang;  /* :619 */
}

void orc_criteria(int n, const float* cmd_x, const float* cmd_yaw, const float* v_fwd_prev,
                  const float* omega_z, float Kp, float Kp_omega, float mass, float inertia,
                  float dt, float* v_delta, float* omega_delta, float* ac_left, float* ac_right) {
  for (int i = 0; i < n; ++i)
    criteria1(cmd_x[i], cmd_yaw[i], v_fwd_prev[i], omega_z[i], Kp, Kp_omega, mass, inertia, dt,
              &v_delta[i], &omega_delta[i], &ac_left[i], &ac_right[i]);
}

/* a3: KAMMA.py:627-628 */
static inline float gamma1f(int32_t sim_steps, float max_sim_steps) {
  float g = 1.0f - ((float)sim_steps / max_sim_steps);
  return g < 0.0f ? 0.0f : g; /* torch.clamp(min=0.0) */
}
void orc_gamma(int n, const int32_t* sim_steps, float max_sim_steps, float* gamma) {
  for (int i = 0; i < n; ++i) gamma[i] = gamma1f(sim_steps[i], max_sim_steps);
}

static inline float clipf(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }

/* a4 + a5(clip) + guiding reward for one arena.
 *   switch: KAMMA.py:632-636, clip :649, guiding :656
 *   blend : KA_DDPG_4D.py:627-631, guiding on the UNCLIPPED policy torque :637 */
static inline void mix1(int mix_mode, int guiding_mode, const float a[4], float ac_left,
                        float ac_right, float gamma1, float rand01, float action_scale, float clip,
                        float torques[4], float wheel_torqs[4], float* guiding) {
  float crit[4] = {ac_left, ac_left, ac_right, ac_right}; /* :624 order fl, rl, fr, rr */
  float d2[4];
  for (int j = 0; j < 4; ++j) {
    float pol = a[j] * action_scale;
    float t;
    if (mix_mode == OIS_MIX_SWITCH) {
      float mask = rand01 < gamma1 ? 1.0f : 0.0f;      /* :634 */
      t = mask * crit[j] + (1.0f - mask) * pol;          /* :635 */
    } else {
      t = ((1.0f - gamma1) * a[j]) * action_scale + gamma1 * crit[j]; /* KA_DDPG_4D.py:631 */
    }
    torques[j] = t;
    wheel_torqs[j] = clipf(t, -clip, clip); /* :649 */
    float d = (guiding_mode == OIS_GUIDE_EXECUTED ? wheel_torqs[j] : pol) - crit[j];
    d2[j] = d * d;
  }
  /* torch.norm over the 4 wheels; summation order fixed by SPEC: (0+1)+(2+3) */
  *guiding = -sqrtf((d2[0] + d2[1]) + (d2[2] + d2[3]));
}

void orc_mix(int n, int mix_mode, int guiding_mode, const float* actions4, const float* ac_left,
             const float* ac_right, const float* gamma1, const float* rand01, float action_scale,
             float clip, float* torques4, float* wheel_torqs4, float* guiding) {
  for (int i = 0; i < n; ++i)
    mix1(mix_mode, guiding_mode, &actions4[4 * i], ac_left[i], ac_right[i], gamma1[i],
         rand01 ? rand01[i] : 0.0f, action_scale, clip, &torques4[4 * i], &wheel_torqs4[4 * i],
         &guiding[i]);
}

/* a7: KAMMA.py:736-764 (env_origins are zero, :224) */
static inline void is_done1(int32_t episode_buf, int max_episode_length, const float q[4],
                            const float pos[3], float bounds, uint8_t* reset, uint8_t* timeout) {
  uint8_t to = episode_buf >= max_episode_length - 1; /* :740-744 */
  const float gvec[3] = {0.0f, 0.0f, -1.0f};          /* :382-384 get_axis_params(-1, 2) */
  float pg[3];
  orc_quat_apply(q, gvec, pg);                         /* :749 */
  uint8_t fallen = pg[2] > 0.0f;                       /* :753 */
  uint8_t oob = (pos[0] < -bounds) | (pos[0] > bounds) | (pos[1] < -bounds) | (pos[1] > bounds); /* :758-759 */
  *timeout = to;
  *reset = (uint8_t)(fallen | to | oob);
}

void orc_is_done(int n, const int32_t* episode_buf, int max_episode_length, const float* quat4,
                 const float* pos3, float bounds, uint8_t* reset, uint8_t* timeout) {
  for (int i = 0; i < n; ++i)
    is_done1(episode_buf[i], max_episode_length, &quat4[4 * i], &pos3[3 * i], bounds, &reset[i], &timeout[i]);
}

typedef struct {
  float lin_acc, ang_acc, rew;
  float comp1, comp2, comp3, rdense, rsparse, robs;
} metrics_out;

/* a8: KAMMA.py:767-822 for one arena */
static inline void metrics1(float v_fwd_prev, float omega_z, float* prev_lin, float* prev_ang,
                            float v_delta, float omega_delta, const float wt[4], float guiding,
                            float gamma2, uint8_t reset, uint8_t timeout, float dt, float w1,
                            float w2, float w3, float sparse_bonus, float sparse_thr,
                            float terminal_reward, float sums[8], metrics_out* o) {
  float lin_acc = (v_fwd_prev - *prev_lin) / dt;   /* :770 */
  float ang_acc = (omega_z - *prev_ang) / dt;      /* :771 */
  *prev_lin = v_fwd_prev;                          /* :774 */
  *prev_ang = omega_z;                             /* :775 */
  float r1 = v_delta * v_delta + 5.0f * (omega_delta * omega_delta);        /* :779 */
  float r2 = lin_acc * lin_acc + ang_acc * ang_acc;                          /* :781 */
  float r3 = (wt[0] * wt[0] + wt[1] * wt[1]) + (wt[2] * wt[2] + wt[3] * wt[3]); /* :783, order by SPEC */
  float c1 = w1 * r1, c2 = w2 * r2, c3 = w3 * r3;
  float rdense = (c1 + c2) + c3;                                             /* :786 */
  float sparse = (fabsf(v_delta) < sparse_thr && fabsf(omega_delta) < sparse_thr) ? sparse_bonus : 0.0f; /* :789-794 */
  float observed = rdense + sparse;                                          /* :795 */
  float rew = (1.0f - gamma2) * observed + gamma2 * guiding;                 /* :798 */
  /* :801  rew += terminalReward * reset_buf * ~timeout_buf ; timeout_buf is an int64
   * tensor, so ~ is the bitwise not: ~0 = -1, ~1 = -2 (reference quirk, kept). */
  rew += (terminal_reward * (float)reset) * (float)(~(int)timeout);
  sums[0] += c1;       /* :803 */
  sums[1] += c2;       /* :804 */
  sums[2] += c3;       /* :805 */
  sums[3] += rdense;   /* :806 */
  sums[4] += sparse;   /* :807 */
  sums[5] += guiding;  /* :808 */
  sums[6] += observed; /* :809 */
  sums[7] += rew;      /* :810 */
  o->lin_acc = lin_acc; o->ang_acc = ang_acc; o->rew = rew;
  o->comp1 = c1; o->comp2 = c2; o->comp3 = c3; o->rdense = rdense; o->rsparse = sparse; o->robs = observed;
}

void orc_metrics(int n, const float* v_fwd_prev, const float* omega_z, float* prev_lin,
                 float* prev_ang, const float* v_delta, const float* omega_delta,
                 const float* wheel_torqs4, const float* guiding, const float* gamma2,
                 const uint8_t* reset, const uint8_t* timeout, float dt, float w1, float w2,
                 float w3, float sparse_bonus, float sparse_thr, float terminal_reward,
                 float* lin_acc, float* ang_acc, float* rew, float* sums8) {
  for (int i = 0; i < n; ++i) {
    metrics_out o;
    metrics1(v_fwd_prev[i], omega_z[i], &prev_lin[i], &prev_ang[i], v_delta[i], omega_delta[i],
             &wheel_torqs4[4 * i], guiding[i], gamma2[i], reset[i], timeout[i], dt, w1, w2, w3,
             sparse_bonus, sparse_thr, terminal_reward, &sums8[8 * i], &o);
    lin_acc[i] = o.lin_acc; ang_acc[i] = o.ang_acc; rew[i] = o.rew;
  }
}

/* a11: KAMMA.py:701-710 (only v_forward_projected is ever read back, :607, :770) */
static inline float forward_projection1(const float q[4], const float v[3]) {
  const float fvec[3] = {1.0f, 0.0f, 0.0f};
  float f[3];
  orc_quat_apply(q, fvec, f);                                        /* :701 */
  float nrm = sqrtf((f[0] * f[0] + f[1] * f[1]) + f[2] * f[2]);      /* :704 torch.norm */
  float fx = f[0] / nrm, fy = f[1] / nrm, fz = f[2] / nrm;
  return (v[0] * fx + v[1] * fy) + v[2] * fz;                        /* :710 torch.sum */
}
void orc_forward_projection(int n, const float* quat4, const float* linvel3, float* v_fwd) {
  for (int i = 0; i < n; ++i) v_fwd[i] = forward_projection1(&quat4[4 * i], &linvel3[3 * i]);
}

/* torch_rand_float(lower, upper, ...) = (upper - lower) * rand + lower
 * (omni.isaac.core.utils.torch.maths, third-party; call sites :424-456, :571-572) */
static inline float rand_float(float lo, float hi, float u) { return (hi - lo) * u + lo; }
/* same with the span (upper - lower) precomputed in double by the host, as Python does */
static inline float rand_span(float lo, float span, float u) { return span * u + lo; }

/* a9 pose + command part, KAMMA.py:422-460 and 490-496 / 569-575 */
void orc_reset_draws(const ois_config* cfg, const float u[8], float pos3[3], float quat4[4],
                     float commands4[4]) {
  float r = cfg->spawn_xy_range;
  pos3[0] = cfg->init_pos[0] + rand_float(-r, r, u[0]); /* :422-424 */
  pos3[1] = cfg->init_pos[1] + rand_float(-r, r, u[1]);
  pos3[2] = cfg->init_pos[2];
  float ang = rand_float(0.0f, 6.28318530717958647692f, u[2]); /* :426 (2*pi as float32) */
  float s, c;
  orc_sincos(ang / 2.0f, &s, &c);                        /* :428, :431 */
  float qz[4] = {c, 0.0f, 0.0f, s};
  orc_quat_mul(qz, cfg->init_quat, quat4);               /* :434 */
  float cx = rand_span(cfg->cmd_x_lo, cfg->cmd_x_span, u[3]);       /* :448 */
  float cy = rand_span(cfg->cmd_y_lo, cfg->cmd_y_span, u[4]);       /* :451 */
  float ch = rand_span(cfg->cmd_yaw_lo, cfg->cmd_yaw_span, u[5]);   /* :454 */
  float keep = sqrtf(cx * cx + cy * cy) > 0.25f ? 1.0f : 0.0f;     /* :458 */
  commands4[0] = cx * keep;
  commands4[1] = cy * keep;
  commands4[2] = 0.0f * keep; /* commands[:,2] is still 0 here (zero-initialised :378) */
  commands4[3] = ch * keep;
  /* sample_velocity_command, randsampling branch :569-575 (u[7] is the discarded omega draw) */
  float x_vel = rand_span(cfg->cmd_x_lo, cfg->cmd_x_span, u[6]);    /* :571 */
  commands4[0] = x_vel > 0.0f ? x_vel : 0.0f;                      /* :575 max(x_vel, 0.0), :495 */
  commands4[2] = 0.0f;                                             /* :574, :496 */
}

/* sample_velocity_command, curriculum branch (KAMMA.py:533-567), for one arena: terrain level,
 * the `fraction` snapshot (KAMMA.py:498), episode_step_counter (KAMMA.py:376, 684: policy steps since
 * post_reset) and one standard normal z standing for torch.normal's draw (mean + std * z). Returns
 * the pair that post_physics_step writes to commands[i, 0] and commands[i, 2] (KAMMA.py:722-725);
 * note the level-0 return order (0.0, x) (KAMMA.py:541). t, and the level-2 ramp, are Python
 * doubles in the reference and doubles here; tensor arithmetic is float32.
 * Where the reference would raise (torch.normal with std < 0: sigma < 0 at level 0 once the r1 sum
 * is below -1.11, Noise < 0 at level 2) the build evaluates mean + std * z all the same. */
/* KAMMA_curr2.py:602-626 (GridSampling): a bimodal normal (sigma 0.05) whose two means move from the centre of the
 * command range to its ends with progress = min(1, sim_steps / max_sim_steps); u_side < 0.5 picks the left mode;
 * torch.normal(mean=Tensor, std=Tensor) = z * std + mean in float32 */
/* BoxSampling bounds (KAMMA_curr1.py:579-591): the range opens from 10 % to all of [lo, hi] around its centre */
void orc_box_bounds(int32_t sim_steps, float max_sim_steps, float lo, float hi, float* xmin, float* xmax) {
  float progress = fminf(1.0f, (float)sim_steps / max_sim_steps);
  float factor = 0.1f + (1.0f - 0.1f) * progress;
  float xc = (lo + hi) / 2.0f;
  *xmin = xc - (xc - lo) * factor;
  *xmax = xc + (hi - xc) * factor;
}

float orc_grid_command(int32_t sim_steps, float max_sim_steps, float lo, float hi, float u_side, float z) {
  float progress = fminf(1.0f, (float)sim_steps / max_sim_steps);
  float xc = (lo + hi) / 2.0f;
  float left = xc - progress * (xc - lo);
  float right = xc + progress * (hi - xc);
  float mean = u_side < 0.5f ? left : right;
  float xv = mean + 0.05f * z;
  return xv > 0.0f ? xv : 0.0f;
}

/* KAMMA_eval.py:545-550 (EvalSampling): x_vel ramps from 1 to 2 m/s with sim_steps / max_sim_steps, omega = 0 */
float orc_eval_command(int32_t sim_steps, float ramp_steps) {
  float progress = fminf(fmaxf((float)sim_steps / ramp_steps, 0.0f), 1.0f);
  float xv = 1.0f + progress * (2.0f - 1.0f);
  return xv > 0.0f ? xv : 0.0f;
}

void orc_level_command(int level, float fraction, int32_t step_counter, float z, double dt, double episode_length_s,
                       float* cmd_x, float* cmd_yaw) {
  if (level == 0) {
    float sigma = 0.01f + 0.09f * fraction;             /* :538 */
    float x = 0.5f + sigma * z;                          /* :539 */
    *cmd_x = 0.0f; *cmd_yaw = x > 0.0f ? x : 0.0f;       /* :541 return 0.0, max(x_vel, 0.0) */
  } else if (level == 1) {
    float freq = 0.01f + 0.09f * fraction;              /* :546 */
    double t = (double)step_counter * dt;                /* :550 */
    float arg = freq * (float)t;                         /* tensor * python float -> float32 */
    float sn = orc_sin(arg);                             /* :551 math.sin */
    float x;
    if (fraction > 0.5f) { float amp = 0.1f + 0.4f * fraction; x = 0.5f + amp * sn; }  /* :548-549 */
    else x = (float)(0.5 + 0.1 * (double)sn);            /* amp stays the Python float 0.1 */
    *cmd_x = x > 0.0f ? x : 0.0f; *cmd_yaw = 0.0f;
  } else if (level == 2) {
    double t = (double)step_counter * dt;                /* :558 */
    float noise = 0.5f * fraction;                       /* :560 */
    float nz = 0.0f + noise * z;                         /* torch.normal(mean=0.0, std=Noise) */
    float x = (float)((double)nz + 0.5 * t / episode_length_s); /* :561, Python doubles */
    *cmd_x = x > 0.0f ? x : 0.0f; *cmd_yaw = 0.0f;
  } else {
    float x = 0.5f + 0.1f * z;                           /* :566 */
    *cmd_x = x > 0.0f ? x : 0.0f; *cmd_yaw = 0.0f;
  }
}
float orc_normal(float u1, float u2) { return normal_det(u1, u2); }

/* ------------------------------------------------------------------ physics ---------- */
typedef struct {
  float dt, inv_dt, gdt, half_dt;
  float im4;         /* 4 / body_mass (mass-splitting factor 4 = contacts per chassis) */
  float imq;         /* 0.25 * im4 (exact): inverse mass seen by the average over the 4 contacts */
  float J4[3];       /* 4 / body_inertia */
  float dt_over_Is, dt_over_Iy, rw, rw_over_Is, rw2_over_Is;
  float beta_inv_dt, vdep, maxw, maxw2, damping;
  float wlim, climit;   /* joint velocity limit and the resistance above it (SPEC 3.3) */
  float cx[4], cy[4], cz; /* wheel centres relative to the composite COM, order fl, rl, fr, rr */
  float inv4wx, inv4wy;   /* 1 / (4 |x|), 1 / (4 |y|) of the wheel centres: footprint-plane fit (SPEC §3.8) */
  int iters;
} derived;

static void derive(const ois_config* c, derived* D) {
  D->dt = c->sim_dt;
  D->inv_dt = 1.0f / c->sim_dt;
  D->gdt = c->gravity * c->sim_dt;
  D->half_dt = 0.5f * c->sim_dt;
  D->im4 = 4.0f / c->body_mass;
  D->imq = 0.25f * D->im4;
  for (int k = 0; k < 3; ++k) D->J4[k] = 4.0f / c->body_inertia[k];
  D->dt_over_Is = c->sim_dt / c->wheel_spin_inertia;
  D->dt_over_Iy = c->sim_dt / c->body_inertia[1];
  D->rw = c->wheel_radius;
  D->rw_over_Is = c->wheel_radius / c->wheel_spin_inertia;
  D->rw2_over_Is = (c->wheel_radius * c->wheel_radius) / c->wheel_spin_inertia;
  D->beta_inv_dt = c->baumgarte * D->inv_dt;
  D->vdep = c->max_depenetration_velocity;
  D->maxw = c->max_angular_velocity;
  D->maxw2 = c->max_angular_velocity * c->max_angular_velocity;
  D->damping = c->wheel_damping;
  D->wlim = c->wheel_limit_velocity;
  D->climit = c->wheel_limit_damping;
  /* env order fl, rl, fr, rr (KAMMA.py:157, 623); URDF: left = +y, front = +x (urdf:112-142) */
  const float sx[4] = {1.0f, -1.0f, 1.0f, -1.0f};
  const float sy[4] = {1.0f, 1.0f, -1.0f, -1.0f};
  for (int w = 0; w < 4; ++w) {
    D->cx[w] = sx[w] * c->wheel_pos[0];
    D->cy[w] = sy[w] * c->wheel_pos[1];
  }
  D->cz = c->wheel_pos[2] - c->com_height;
  D->inv4wx = 1.0f / (4.0f * c->wheel_pos[0]);
  D->inv4wy = 1.0f / (4.0f * c->wheel_pos[1]);
  D->iters = c->solver_iterations;
}

static inline void rotmat(const float q[4], float R[9]) {
  float qw = q[0], qx = q[1], qy = q[2], qz = q[3];
  float x2 = qx + qx, y2 = qy + qy, z2 = qz + qz;
  float xx = qx * x2, yy = qy * y2, zz = qz * z2;
  float xy = qx * y2, xz = qx * z2, yz = qy * z2;
  float wx = qw * x2, wy = qw * y2, wz = qw * z2;
  R[0] = 1.0f - (yy + zz); R[1] = xy - wz;          R[2] = xz + wy;
  R[3] = xy + wz;          R[4] = 1.0f - (xx + zz); R[5] = yz - wx;
  R[6] = xz - wy;          R[7] = yz + wx;          R[8] = 1.0f - (xx + yy);
}
/* out = R^T a */
static inline void rotT(const float R[9], const float a[3], float o[3]) {
  o[0] = fmaf(R[6], a[2], fmaf(R[3], a[1], R[0] * a[0]));
  o[1] = fmaf(R[7], a[2], fmaf(R[4], a[1], R[1] * a[0]));
  o[2] = fmaf(R[8], a[2], fmaf(R[5], a[1], R[2] * a[0]));
}
/* out = R a */
static inline void rot(const float R[9], const float a[3], float o[3]) {
  o[0] = fmaf(R[2], a[2], fmaf(R[1], a[1], R[0] * a[0]));
  o[1] = fmaf(R[5], a[2], fmaf(R[4], a[1], R[3] * a[0]));
  o[2] = fmaf(R[8], a[2], fmaf(R[7], a[1], R[6] * a[0]));
}

/* One physics substep (docs/SPEC.md §3). spin[] = absolute wheel spin about the body y
 * axis (joint velocity + chassis w_y). Mass-splitting Jacobi across the 4 wheel contacts,
 * Gauss-Seidel across the 3 rows (normal, longitudinal, lateral) of each contact.
 * v is the WORLD linear velocity, wb the BODY-frame angular velocity: a torque-free body
 * with gyroscopic forces disabled (KAMMA.py:119) keeps its world w constant, and
 * d(R^T w)/dt = -w_b x w_b = 0, so w_b is carried from substep to substep unchanged. */
/* Reciprocal cache carried from substep to substep inside one policy step (SPEC §3.6): the
 * row effective masses 1/k and 1/s change by O(1e-3) between substeps, so after the first
 * substep (exact IEEE division) they are refreshed by ONE Newton step from the previous
 * value — multiplications and fmas only, bit-reproducible on CPU and GPU. The effective mass
 * is only the gain of the projected Gauss-Seidel update: its 1e-6 relative error does not
 * move the fixed point of the solve. */
typedef struct {
  float mn[4], m1[4], m2[4], inv_s;
} rcache;

static inline float newton_rcp(float m, float k) { /* m ~ 1/k -> m (2 - k m) */
  float e = fmaf(-k, m, 1.0f);
  return fmaf(m, e, m);
}
static inline float newton_rsqrt(float r, float s2) { /* r ~ 1/sqrt(s2) -> r (1.5 - 0.5 s2 r^2) */
  float hr = (0.5f * s2) * r;
  float e = fmaf(-hr, r, 1.5f);
  return r * e;
}

/* ---- height-field terrain (docs/SPEC.md §3.8; build-defined like the rest of the physics) ---- */
typedef struct {
  const int16_t* h; /* [rows, cols] raw heights */
  int rows, cols;
  float inv_hs, hs, vs, border;
  int stride; /* ois_terrain.footprint_stride: 1 = footprints sampled every substep, 2 = every second one (SPEC 3.8) */
} terrain_ctx;

/* ground height at world (x, y): the sampling rule of get_heights (KAMMA.py:882-893: pixel =
 * trunc((x + border) / horizontal_scale) clipped to [0, size-2], min of the pixel and its
 * (+1,+1) neighbour) with the division replaced by a multiplication with 1/horizontal_scale */
static inline float terrain_height(const terrain_ctx* T, float x, float y) {
  float fx = (x + T->border) * T->inv_hs, fy = (y + T->border) * T->inv_hs;
  fx = fminf(fmaxf(fx, 0.0f), (float)(T->rows - 2));
  fy = fminf(fmaxf(fy, 0.0f), (float)(T->cols - 2));
  int px = (int)fx, py = (int)fy;
  int16_t h1 = T->h[(size_t)px * T->cols + py], h2 = T->h[(size_t)(px + 1) * T->cols + (py + 1)];
  return (float)(h1 < h2 ? h1 : h2) * T->vs;
}

/* Footprints of the 4 wheels: world xy under the wheel centres (arena-local), the field height
 * there (tile origin subtracted) and the xy part of the body x / y axes at sampling time. */
typedef struct {
  float wx[4], wy[4], h[4];
  float r00, r10, r01, r11;
} footprints;
/* what a substep hands to the next one on a height field: the stored sample and, for footprint_stride 2, the plane fitted to it */
typedef struct {
  footprints fp;
  float n[3], off[4];
} terrain_carry;

static void terrain_footprints(const derived* D, const terrain_ctx* T, const float origin[3], const float p[3],
                               const float R[9], footprints* F) {
  for (int k = 0; k < 4; ++k) {
    F->wx[k] = fmaf(R[2], D->cz, fmaf(R[1], D->cy[k], fmaf(R[0], D->cx[k], p[0])));
    F->wy[k] = fmaf(R[5], D->cz, fmaf(R[4], D->cy[k], fmaf(R[3], D->cx[k], p[1])));
    F->h[k] = terrain_height(T, F->wx[k] + origin[0], F->wy[k] + origin[1]) - origin[2];
  }
  F->r00 = R[0]; F->r10 = R[3]; F->r01 = R[1]; F->r11 = R[4];
}

/* The wheels stand on the plane through their 4 footprints: unit normal n from the front-rear
 * and left-right footprint differences, off[k] = n . footprint_k (plane offset seen by wheel k). */
static void terrain_plane(const derived* D, const footprints* F, float n[3], float off[4]) {
  const float* h = F->h;
  float g1 = ((h[0] + h[2]) - (h[1] + h[3])) * D->inv4wx; /* front - rear (order fl, rl, fr, rr) */
  float g2 = ((h[0] + h[1]) - (h[2] + h[3])) * D->inv4wy; /* left - right */
  /* e1 = (R00, R10, g1), e2 = (R01, R11, g2): n ~ e1 x e2 */
  float nx = fmaf(F->r10, g2, -(g1 * F->r11));
  float ny = fmaf(g1, F->r01, -(F->r00 * g2));
  float nz = fmaf(F->r00, F->r11, -(F->r10 * F->r01));
  float nn = fmaxf(fmaf(nz, nz, fmaf(ny, ny, nx * nx)), 1e-12f);
  float inv = 1.0f / sqrtf(nn);
  if (nz < 0.0f) inv = -inv; /* rolled over (body z points down): the ground's normal still points up */
  n[0] = nx * inv; n[1] = ny * inv; n[2] = nz * inv;
  for (int k = 0; k < 4; ++k) off[k] = fmaf(n[2], h[k], fmaf(n[1], F->wy[k], n[0] * F->wx[k]));
}

static void substep(const derived* D, float mu, const float n_in[3], const float tau[4], float p[3],
                    float q[4], float v[3], float wb[3], float spin[4], float ln[4], float l1[4],
                    float l2[4], rcache* C, int sub, const terrain_ctx* TR, const float* origin, terrain_carry* TC) {
  const int exact = sub == 0;
  /* footprint_stride 2 (SPEC 3.8): footprints are sampled at the start of EVEN substeps only. An odd substep fits the plane to
   * the sample of the substep before it; an even substep >= 2 keeps that plane (n and the offsets n . footprint_k). */
  const int stride2 = TR && TR->stride == 2;
  const int sample = TR && (!stride2 || (sub & 1) == 0);
  const int keep_plane = stride2 && !exact && (sub & 1) == 0;
  float R[9];
  rotmat(q, R);
  float vb[3], nb[3];
  float n[3] = {n_in[0], n_in[1], n_in[2]}, off[4] = {0.0f, 0.0f, 0.0f, 0.0f};
  footprints cur;
  if (TR) {
    /* the plane is fitted to the footprints sampled one substep EARLIER (the first substep of a
     * policy step samples its own): on the GPU the 8 gathers of this substep are then in flight
     * during the whole solve instead of stalling it (SPEC §3.8). A wheel moves <= 2 cm per substep. */
    if (sample) terrain_footprints(D, TR, origin, p, R, &cur);
    if (exact) TC->fp = cur;
    if (keep_plane) {
      for (int k = 0; k < 3; ++k) n[k] = TC->n[k];
      for (int k = 0; k < 4; ++k) off[k] = TC->off[k];
    } else {
      terrain_plane(D, &TC->fp, n, off);
      for (int k = 0; k < 3; ++k) TC->n[k] = n[k];
      for (int k = 0; k < 4; ++k) TC->off[k] = off[k];
    }
  }
  rotT(R, v, vb);
  rotT(R, n, nb);
  /* gravity (world -z) in the body frame */
  vb[0] = fmaf(-D->gdt, R[6], vb[0]);
  vb[1] = fmaf(-D->gdt, R[7], vb[1]);
  vb[2] = fmaf(-D->gdt, R[8], vb[2]);
  /* wheel drive + viscous joint damping, reaction on the chassis */
  float T[4];
  for (int k = 0; k < 4; ++k) {
    float jv = spin[k] - wb[1];
    float over = jv - fminf(fmaxf(jv, -D->wlim), D->wlim); /* signed excess over the joint velocity limit */
    T[k] = fmaf(-D->climit, over, fmaf(-D->damping, jv, tau[k]));
    spin[k] = fmaf(D->dt_over_Is, T[k], spin[k]);
  }
  float Tsum = (T[0] + T[1]) + (T[2] + T[3]);
  wb[1] = fmaf(-D->dt_over_Iy, Tsum, wb[1]);

  /* contact frame shared by the four wheels (common axle direction = body y) */
  float s2 = fmaf(nb[2], nb[2], nb[0] * nb[0]);
  s2 = fmaxf(s2, 1e-12f);
  /* on a height field exact reciprocals in every substep, the ones that keep the plane included: a robot that tumbles over the
   * field passes through poses (on its side: n_b along the axle) where the contact points, and with them the effective masses,
   * move by a wheel radius within a substep, and a Newton refresh from the previous value then returns garbage that is
   * finite (round 4: 1 000 m/s launches in 1 of ~10^4 arenas per 10 steps with the refresh on the plane-keeping substeps) */
  const int rex = exact || TR != NULL;
  float inv_s = rex ? 1.0f / sqrtf(s2) : newton_rsqrt(C->inv_s, s2);
  C->inv_s = inv_s;
  float dx = -(nb[0] * inv_s), dz = -(nb[2] * inv_s);
  float t1x = -dz, t1z = dx;
  float t2x = nb[1] * t1z;
  float t2y = fmaf(nb[2], t1x, -(nb[0] * t1z));
  float t2z = -(nb[1] * t1x);
  float np = fmaf(n[2], p[2], fmaf(n[1], p[1], n[0] * p[0]));

  float b[4], an[4][3], a1[4][3], a2[4][3], Jan[4][3], Ja1[4][3], Ja2[4][3];
  float *mn = C->mn, *m1 = C->m1, *m2 = C->m2;
  for (int k = 0; k < 4; ++k) {
    float cx = D->cx[k], cy = D->cy[k], cz = D->cz;
    float rx = fmaf(D->rw, dx, cx), ry = cy, rz = fmaf(D->rw, dz, cz);
    float gap = (TR ? np - off[k] : np) + fmaf(nb[2], rz, fmaf(nb[1], ry, nb[0] * rx));
    /* gap >= 0: -gap/dt (speculative contact); gap < 0: min(-beta gap/dt, vdep). With 0 < beta < 1 the
     * first is the smaller of the two products exactly when gap >= 0, so both cases are one min3 */
    b[k] = fminf(fminf(-(gap * D->inv_dt), -(gap * D->beta_inv_dt)), D->vdep);
    an[k][0] = fmaf(ry, nb[2], -(rz * nb[1]));
    an[k][1] = fmaf(rz, nb[0], -(rx * nb[2]));
    an[k][2] = fmaf(rx, nb[1], -(ry * nb[0]));
    a1[k][0] = cy * t1z;
    a1[k][1] = fmaf(cz, t1x, -(cx * t1z));
    a1[k][2] = -(cy * t1x);
    a2[k][0] = fmaf(ry, t2z, -(rz * t2y));
    a2[k][1] = fmaf(rz, t2x, -(rx * t2z));
    a2[k][2] = fmaf(rx, t2y, -(ry * t2x));
    for (int j = 0; j < 3; ++j) {
      Jan[k][j] = D->J4[j] * an[k][j];
      Ja1[k][j] = D->J4[j] * a1[k][j];
      Ja2[k][j] = D->J4[j] * a2[k][j];
    }
    float kn = fmaf(an[k][2], Jan[k][2], fmaf(an[k][1], Jan[k][1], fmaf(an[k][0], Jan[k][0], D->im4)));
    float k1 = fmaf(a1[k][2], Ja1[k][2], fmaf(a1[k][1], Ja1[k][1], fmaf(a1[k][0], Ja1[k][0], D->im4))) + D->rw2_over_Is;
    float k2 = fmaf(a2[k][2], Ja2[k][2], fmaf(a2[k][1], Ja2[k][1], fmaf(a2[k][0], Ja2[k][0], D->im4)));
    mn[k] = rex ? 1.0f / kn : newton_rcp(mn[k], kn);
    m1[k] = rex ? 1.0f / k1 : newton_rcp(m1[k], k1);
    m2[k] = rex ? 1.0f / k2 : newton_rcp(m2[k], k2);
  }
  /* Chassis linear velocity: the contact frame (n, t1, t2) is orthonormal and the split inverse
   * mass is a scalar, so an impulse along one row changes only that row's own linear velocity
   * component. The solve therefore carries the three PROJECTIONS (un, u1, u2) of v_b on the frame
   * instead of v_b: a row reads its component, the averaging pass adds (1/m) * (sum of the pass's
   * impulses of that row), and v_b is rebuilt from the frame once, after the last pass (SPEC 3.5). */
  float un = fmaf(nb[2], vb[2], fmaf(nb[1], vb[1], nb[0] * vb[0]));
  float u1 = fmaf(t1z, vb[2], t1x * vb[0]);
  float u2 = fmaf(t2z, vb[2], fmaf(t2y, vb[1], t2x * vb[0]));

  /* warm start: re-apply the impulses accumulated in the previous substep (they persist
   * across policy steps in OIS_BUF_LAMBDA_*), as one more mass-split averaging pass */
  {
    float wl[4][3];
    for (int k = 0; k < 4; ++k) {
      for (int j = 0; j < 3; ++j) wl[k][j] = fmaf(l2[k], Ja2[k][j], fmaf(l1[k], Ja1[k][j], fmaf(ln[k], Jan[k][j], wb[j])));
      spin[k] = fmaf(-D->rw_over_Is, l1[k], spin[k]);
    }
    float Sn = (ln[0] + ln[1]) + (ln[2] + ln[3]);
    float S1 = (l1[0] + l1[1]) + (l1[2] + l1[3]);
    float S2 = (l2[0] + l2[1]) + (l2[2] + l2[3]);
    un = fmaf(D->imq, Sn, un); u1 = fmaf(D->imq, S1, u1); u2 = fmaf(D->imq, S2, u2);
    for (int j = 0; j < 3; ++j) wb[j] = 0.25f * ((wl[0][j] + wl[1][j]) + (wl[2][j] + wl[3][j]));
  }
  for (int it = 0; it < D->iters; ++it) {
    float wl[4][3], dn[4], d1[4], d2[4];
    for (int k = 0; k < 4; ++k) {
      float ox = wb[0], oy = wb[1], oz = wb[2];
      /* normal row */
      float u = fmaf(an[k][2], oz, fmaf(an[k][1], oy, fmaf(an[k][0], ox, un)));
      float nw = fmaxf(fmaf(b[k] - u, mn[k], ln[k]), 0.0f);
      float dl = nw - ln[k];
      ln[k] = nw;
      dn[k] = dl;
      ox = fmaf(dl, Jan[k][0], ox); oy = fmaf(dl, Jan[k][1], oy); oz = fmaf(dl, Jan[k][2], oz);
      float lim = mu * ln[k];
      /* longitudinal row (couples the wheel spin) */
      u = fmaf(-D->rw, spin[k], fmaf(a1[k][2], oz, fmaf(a1[k][1], oy, fmaf(a1[k][0], ox, u1))));
      nw = fminf(fmaxf(fmaf(-u, m1[k], l1[k]), -lim), lim);
      dl = nw - l1[k];
      l1[k] = nw;
      d1[k] = dl;
      ox = fmaf(dl, Ja1[k][0], ox); oy = fmaf(dl, Ja1[k][1], oy); oz = fmaf(dl, Ja1[k][2], oz);
      spin[k] = fmaf(-D->rw_over_Is, dl, spin[k]);
      /* lateral row */
      u = fmaf(a2[k][2], oz, fmaf(a2[k][1], oy, fmaf(a2[k][0], ox, u2)));
      nw = fminf(fmaxf(fmaf(-u, m2[k], l2[k]), -lim), lim);
      dl = nw - l2[k];
      l2[k] = nw;
      d2[k] = dl;
      ox = fmaf(dl, Ja2[k][0], ox); oy = fmaf(dl, Ja2[k][1], oy); oz = fmaf(dl, Ja2[k][2], oz);
      wl[k][0] = ox; wl[k][1] = oy; wl[k][2] = oz;
    }
    float Sn = (dn[0] + dn[1]) + (dn[2] + dn[3]);
    float S1 = (d1[0] + d1[1]) + (d1[2] + d1[3]);
    float S2 = (d2[0] + d2[1]) + (d2[2] + d2[3]);
    un = fmaf(D->imq, Sn, un); u1 = fmaf(D->imq, S1, u1); u2 = fmaf(D->imq, S2, u2);
    for (int j = 0; j < 3; ++j) wb[j] = 0.25f * ((wl[0][j] + wl[1][j]) + (wl[2][j] + wl[3][j]));
  }
  /* rebuild v_b from its projections on the (orthonormal) contact frame; t1 has no y component */
  vb[0] = fmaf(t2x, u2, fmaf(t1x, u1, nb[0] * un));
  vb[1] = fmaf(t2y, u2, nb[1] * un);
  vb[2] = fmaf(t2z, u2, fmaf(t1z, u1, nb[2] * un));
  /* back to the world frame, velocity caps (origin_v18.py:65-70), integrate */
  rot(R, vb, v);
  float w2 = fmaf(wb[2], wb[2], fmaf(wb[1], wb[1], wb[0] * wb[0]));
  if (w2 > D->maxw2) {
    float sc = D->maxw / sqrtf(w2);
    wb[0] *= sc; wb[1] *= sc; wb[2] *= sc;
  }
  for (int k = 0; k < 4; ++k) spin[k] = fminf(fmaxf(spin[k], -D->maxw), D->maxw);
  p[0] = fmaf(D->dt, v[0], p[0]);
  p[1] = fmaf(D->dt, v[1], p[1]);
  p[2] = fmaf(D->dt, v[2], p[2]);
  /* q += dt/2 * q (x) (0, w_b), then renormalise */
  float qw = q[0], qx = q[1], qy = q[2], qz = q[3], h = D->half_dt;
  float nqw = fmaf(-h, fmaf(qz, wb[2], fmaf(qy, wb[1], qx * wb[0])), qw);
  float nqx = fmaf(h, fmaf(-qz, wb[1], fmaf(qy, wb[2], qw * wb[0])), qx);
  float nqy = fmaf(h, fmaf(-qx, wb[2], fmaf(qz, wb[0], qw * wb[1])), qy);
  float nqz = fmaf(h, fmaf(-qy, wb[0], fmaf(qx, wb[1], qw * wb[2])), qz);
  float nn = fmaf(nqz, nqz, fmaf(nqy, nqy, fmaf(nqx, nqx, nqw * nqw)));
  /* |q|^2 = 1 + O((w dt)^2): one Newton step of 1/sqrt is exact to float precision; the
   * exact form only when the norm has drifted (extreme spin) */
  float inv = fmaf(-0.5f, nn, 1.5f);
  if (fabsf(nn - 1.0f) > 1e-3f) inv = 1.0f / sqrtf(nn);
  q[0] = nqw * inv; q[1] = nqx * inv; q[2] = nqy * inv; q[3] = nqz * inv;
  if (sample) TC->fp = cur;
}

/* w is the WORLD angular velocity here (converted to the body frame and back) */
void orc_substep(const ois_config* cfg, float mu, const float n_world[3], const float tau[4],
                 float p[3], float q[4], float v[3], float w[3], float spin[4], float lam_n[4],
                 float lam_l[4], float lam_t[4]) {
  derived D;
  derive(cfg, &D);
  float R[9], wb[3];
  rotmat(q, R);
  rotT(R, w, wb);
  rcache C;
  substep(&D, mu, n_world, tau, p, q, v, wb, spin, lam_n, lam_l, lam_t, &C, 0, NULL, NULL, NULL);
  rotmat(q, R);
  rot(R, wb, w);
}

/* ------------------------------------------------------------------ environment ------ */
#define GROUP 16          /* arenas per reduction group (= one wave64 of the HIP kernel) */
#define PART_STRIDE 32    /* floats per group partial record */
#define CHUNK_GROUPS 64   /* groups per level-1 chunk of the logging reductions (= 1024 arenas) */
/* partial-record slots: 0..15 as OIS_EI_COUNT..OBSERVED_OVER_FINAL (sums over reset arenas),
 * 16 gamma1, 17 gamma2, 20..22 eval sums, 24 return^2 sum, 25 return min, 26 return max, 27 length sum */
enum { P_G1 = 16, P_G2 = 17, P_SM = 20, P_TE = 21, P_DV = 22, P_RET2 = 24, P_RMIN = 25, P_RMAX = 26, P_LEN = 27 };
enum { ST_COUNT = 0, ST_RET = 1, ST_RET2 = 2, ST_RMIN = 3, ST_RMAX = 4, ST_LEN = 5, ST_STRIDE = 8 };

struct orc_env {
  ois_config cfg;
  derived D;
  int N, G;
  int threads;
  int reset_done;
  int old_torques_zeroed; /* KAMMA_eval.py:470: the last step's reset_idx zeroed old_torques of EVERY arena */
  uint64_t seed;
  double steps_done;
  /* buffers, same ids/shapes as the product */
  float *obs, *rew, *wheel_torqs, *pose, *quat, *linvel, *angvel, *wheel_vel, *commands, *sums_a, *sums_b, *sums_c, *params;
  uint16_t *lam_n, *lam_l, *lam_t; /* contact impulse cache (warm start), binary16 [N,4] each */
  float* cmd_fraction;          /* [N] self.fraction (KAMMA.py:498), OIS_CMD_LEVEL only */
  uint8_t *done, *timeout;
  int32_t* counters;
  float episode_info[OIS_EPISODE_INFO_LEN];
  float env0_diag[OIS_ENV0_DIAG_LEN];
  float* partials;  /* [G][PART_STRIDE] */
  int32_t* stamps;  /* [G] */
  double* stat_acc; /* [G][ST_STRIDE] finished-episode accumulators, double */
  /* height-field terrain (mirror of ois_terrain_attach): own copies of the field and the table */
  int terrain_on;
  ois_terrain tcfg;
  terrain_ctx terrain;
  int16_t* theights;
  float* terrain_origins; /* [levels, types, 3] */
  float* env_origins;     /* [N,4] */
  int32_t* terrain_idx;   /* [N,4] level, type */
  /* FIFO transition memory (mirror of ois_replay_*) */
  float *rp_states, *rp_actions, *rp_rewards, *rp_next;
  uint8_t* rp_done;
  int rp_size, rp_head, rp_filled;
  orc_test_hooks hooks; /* fixture replay (orc_test_set_hooks); all NULL in normal use */
};

static void* zalloc(size_t n) { return calloc(n ? n : 1, 1); }

int orc_create(const ois_config* cfg, orc_env** out) {
  if (!cfg || !out) return fail(OIS_EINVAL, "null argument");
  if (cfg->struct_bytes != sizeof(ois_config)) return fail(OIS_EINVAL, "ois_config.struct_bytes mismatch");
  if (cfg->num_envs <= 0) return fail(OIS_EINVAL, "num_envs must be > 0");
  if (cfg->action_dim != 4 && cfg->action_dim != 2 && cfg->action_dim != 1) return fail(OIS_EINVAL, "action_dim must be 4, 2 or 1");
  if (cfg->solver_iterations < 1 || cfg->solver_iterations > 16) return fail(OIS_EINVAL, "solver_iterations out of range");
  if (!(cfg->baumgarte > 0.0f && cfg->baumgarte < 1.0f)) return fail(OIS_EINVAL, "baumgarte must be in (0, 1)");
  if (!(cfg->wheel_damping >= 0.0f) || !(cfg->wheel_limit_damping >= 0.0f) || !(cfg->wheel_limit_velocity >= 0.0f))
    return fail(OIS_EINVAL, "wheel joint resistance constants must be >= 0");
  if (!((cfg->wheel_damping + cfg->wheel_limit_damping) * cfg->sim_dt / cfg->wheel_spin_inertia < 2.0f))
    return fail(OIS_EINVAL, "(wheel_damping + wheel_limit_damping) * sim_dt / wheel_spin_inertia must be < 2 (explicit integration)");
  if (!(cfg->max_depenetration_velocity >= 0.0f)) return fail(OIS_EINVAL, "max_depenetration_velocity must be >= 0");
  if (!(cfg->friction >= 0.0f) || !(cfg->friction < 1e6f)) return fail(OIS_EINVAL, "friction must be >= 0 (finite)");
  orc_env* e = (orc_env*)zalloc(sizeof *e);
  if (!e) return fail(OIS_ENOMEM, "host alloc");
  e->cfg = *cfg;
  derive(cfg, &e->D);
  int N = e->N = cfg->num_envs;
  int G = e->G = (N + GROUP - 1) / GROUP;
  e->threads = 1;
  size_t f4 = (size_t)N * 4 * sizeof(float);
  e->obs = zalloc((size_t)N * 6 * sizeof(float));
  e->rew = zalloc((size_t)N * sizeof(float));
  e->wheel_torqs = zalloc(f4); e->pose = zalloc(f4); e->quat = zalloc(f4); e->linvel = zalloc(f4);
  e->angvel = zalloc(f4); e->wheel_vel = zalloc(f4); e->commands = zalloc(f4);
  e->sums_a = zalloc(f4); e->sums_b = zalloc(f4); e->sums_c = zalloc(f4); e->params = zalloc(f4);
  e->lam_n = zalloc(f4 / 2); e->lam_l = zalloc(f4 / 2); e->lam_t = zalloc(f4 / 2);
  e->cmd_fraction = zalloc((size_t)N * sizeof(float));
  e->done = zalloc((size_t)N); e->timeout = zalloc((size_t)N);
  e->counters = zalloc((size_t)N * 4 * sizeof(int32_t));
  e->env_origins = zalloc(f4);
  e->terrain_idx = zalloc((size_t)N * 4 * sizeof(int32_t));
  e->partials = zalloc((size_t)G * PART_STRIDE * sizeof(float));
  e->stamps = zalloc((size_t)G * sizeof(int32_t));
  e->stat_acc = zalloc((size_t)G * ST_STRIDE * sizeof(double));
  for (int i = 0; i < N; ++i) {
    e->params[4 * i + 0] = cfg->friction;
    e->params[4 * i + 1] = 0.0f; e->params[4 * i + 2] = 0.0f; e->params[4 * i + 3] = 1.0f;
  }
  *out = e;
  return OIS_OK;
}

void orc_destroy(orc_env* e) {
  if (!e) return;
  free(e->obs); free(e->rew); free(e->wheel_torqs); free(e->pose); free(e->quat); free(e->linvel);
  free(e->angvel); free(e->wheel_vel); free(e->commands); free(e->sums_a); free(e->sums_b); free(e->sums_c);
  free(e->params); free(e->lam_n); free(e->lam_l); free(e->lam_t); free(e->done); free(e->timeout); free(e->counters); free(e->partials); free(e->stamps);
  free(e->stat_acc);
  free(e->rp_states); free(e->rp_actions); free(e->rp_rewards); free(e->rp_next); free(e->rp_done);
  free(e->theights); free(e->terrain_origins); free(e->env_origins); free(e->terrain_idx);
  free(e->cmd_fraction);
  free(e);
}

int orc_set_threads(orc_env* e, int threads) {
  if (!e || threads < 1) return fail(OIS_EINVAL, "threads must be >= 1");
#ifndef _OPENMP
  if (threads != 1) return fail(OIS_EINVAL, "oracle built without OpenMP");
#endif
  e->threads = threads;
  return OIS_OK;
}

int orc_test_set_hooks(orc_env* e, const orc_test_hooks* h) {
  if (!e) return fail(OIS_EINVAL, "null env");
  if (h) e->hooks = *h; else memset(&e->hooks, 0, sizeof e->hooks);
  return OIS_OK;
}

int orc_buffer(orc_env* e, int id, void** ptr, size_t* bytes) {
  if (!e || !ptr || !bytes) return fail(OIS_EINVAL, "null argument");
  size_t N = (size_t)e->N, f4 = N * 4 * sizeof(float);
  switch (id) {
    case OIS_BUF_OBS: *ptr = e->obs; *bytes = N * 6 * sizeof(float); break;
    case OIS_BUF_REW: *ptr = e->rew; *bytes = N * sizeof(float); break;
    case OIS_BUF_DONE: *ptr = e->done; *bytes = N; break;
    case OIS_BUF_TIMEOUT: *ptr = e->timeout; *bytes = N; break;
    case OIS_BUF_WHEEL_TORQS: *ptr = e->wheel_torqs; *bytes = f4; break;
    case OIS_BUF_POSE: *ptr = e->pose; *bytes = f4; break;
    case OIS_BUF_QUAT: *ptr = e->quat; *bytes = f4; break;
    case OIS_BUF_LINVEL: *ptr = e->linvel; *bytes = f4; break;
    case OIS_BUF_ANGVEL: *ptr = e->angvel; *bytes = f4; break;
    case OIS_BUF_WHEEL_VEL: *ptr = e->wheel_vel; *bytes = f4; break;
    case OIS_BUF_COMMANDS: *ptr = e->commands; *bytes = f4; break;
    case OIS_BUF_COUNTERS: *ptr = e->counters; *bytes = N * 4 * sizeof(int32_t); break;
    case OIS_BUF_SUMS_A: *ptr = e->sums_a; *bytes = f4; break;
    case OIS_BUF_SUMS_B: *ptr = e->sums_b; *bytes = f4; break;
    case OIS_BUF_SUMS_C: *ptr = e->sums_c; *bytes = f4; break;
    case OIS_BUF_PARAMS: *ptr = e->params; *bytes = f4; break;
    case OIS_BUF_LAMBDA_N: *ptr = e->lam_n; *bytes = f4 / 2; break;
    case OIS_BUF_LAMBDA_L: *ptr = e->lam_l; *bytes = f4 / 2; break;
    case OIS_BUF_LAMBDA_T: *ptr = e->lam_t; *bytes = f4 / 2; break;
    case OIS_BUF_ENV_ORIGINS: *ptr = e->env_origins; *bytes = f4; break;
    case OIS_BUF_TERRAIN_IDX: *ptr = e->terrain_idx; *bytes = N * 4 * sizeof(int32_t); break;
    case OIS_BUF_CMD_FRACTION: *ptr = e->cmd_fraction; *bytes = N * sizeof(float); break;
    case OIS_BUF_EPISODE_INFO: *ptr = e->episode_info; *bytes = sizeof e->episode_info; break;
    case OIS_BUF_ENV0_DIAG: *ptr = e->env0_diag; *bytes = sizeof e->env0_diag; break;
    default: return fail(OIS_EINVAL, "unknown buffer id");
  }
  return OIS_OK;
}

/* commands follow the per-level generators: config switch, or an attached terrain with curriculum */
static int level_mode(const orc_env* e) {
  return e->cfg.cmd_sampler == OIS_CMD_LEVEL || (e->terrain_on && e->tcfg.curriculum);
}

/* per-arena values contributed to the group partial record when the arena resets */
typedef struct {
  float v[16];  /* slots 0..15 */
  float sm, te, dv;
  float ret, len;
} reset_contrib;

/* the per-arena values reset_idx contributes to extras["episode"] (KAMMA.py:470-485): slot 0 the count, 1..8 the episode
 * sums, 9..15 the ratio sums L * (100 * (1 - gamma2) * S_k / S_final) (gamma2 itself for the guiding one, :475) */
void orc_reset_ratios(const float S[8], float gamma2, float L, float v[16]) {
  float Sf = S[7];
  float hg = 100.0f * (1.0f - gamma2); /* hundred * (one - gamma2) :470 */
  float gg = 100.0f * gamma2;          /* hundred * gamma2         :475 */
  v[OIS_EI_COUNT] = 1.0f;
  for (int k = 0; k < 8; ++k) v[OIS_EI_R1 + k] = S[k];
  v[OIS_EI_R1_OVER_FINAL] = L * ((hg * S[0]) / Sf);       /* :470 */
  v[OIS_EI_R2_OVER_FINAL] = L * ((hg * S[1]) / Sf);       /* :471 */
  v[OIS_EI_R3_OVER_FINAL] = L * ((hg * S[2]) / Sf);       /* :472 */
  v[OIS_EI_DENSE_OVER_FINAL] = L * ((hg * S[3]) / Sf);    /* :473 */
  v[OIS_EI_SPARSE_OVER_FINAL] = L * ((hg * S[4]) / Sf);   /* :474 */
  v[OIS_EI_GUIDING_OVER_FINAL] = L * ((gg * S[5]) / Sf);  /* :475 */
  v[OIS_EI_OBSERVED_OVER_FINAL] = L * ((hg * S[6]) / Sf); /* :476 */
}

/* reset_idx for ONE arena (KAMMA.py:417-498). Fills the contribution to extras, rewrites
 * pose / velocities / commands / counters / sums. step_ctr = sim_steps value used as RNG
 * counter. gamma2 = current reward assist (:467-476). */
static void reset_arena(orc_env* e, int i, uint32_t step_ctr, float gamma2, reset_contrib* rc) {
  const ois_config* c = &e->cfg;
  float* sa = &e->sums_a[4 * i];
  float* sb = &e->sums_b[4 * i];
  float* sc = &e->sums_c[4 * i];
  int32_t* cnt = &e->counters[4 * i];
  float S[8] = {sa[0], sa[1], sa[2], sa[3], sb[0], sb[1], sb[2], sb[3]};
  float Sf = S[7];
  orc_reset_ratios(S, gamma2, c->episode_length_s, rc->v);
  rc->sm = sc[0]; rc->te = sc[1]; rc->dv = sc[2];
  rc->ret = Sf;
  rc->len = (float)cnt[0];
  if (e->terrain_on && e->tcfg.curriculum && step_ctr != 0u) {
    /* update_SI_level (KAMMA.py:502-529; skipped while not init_done, :504): one level down
     * when the finished episode's r1 sum is below threshold_low, one up above threshold_high.
     * (The reference writes through a chained index, terrain_levels[env_ids][mask] = ..., which
     * assigns to a temporary; the evident intent is implemented.) */
    int32_t* ti = &e->terrain_idx[4 * i];
    int lvl = ti[0];
    if (S[0] < e->tcfg.threshold_low) lvl = lvl - 1 < 0 ? 0 : lvl - 1;
    if (S[0] > e->tcfg.threshold_high) lvl = lvl + 1 > e->tcfg.max_level ? e->tcfg.max_level : lvl + 1;
    ti[0] = lvl;
    const float* o = &e->terrain_origins[((size_t)lvl * e->tcfg.num_types + ti[1]) * 3];
    float* eo = &e->env_origins[4 * i];
    eo[0] = o[0]; eo[1] = o[1]; eo[2] = o[2]; /* :529 */
  }
  /* :485 zero the sums of the reset arena */
  for (int k = 0; k < 4; ++k) { sa[k] = 0.0f; sb[k] = 0.0f; }
  sc[0] = sc[1] = sc[2] = 0.0f;
  sc[3] = 0.0f; /* old_torques of this arena are zero (KAMMA_eval.py:470; the other arenas': step_all) */

  float u[8];
  int64_t gid = c->arena_offset + i;
  if (e->hooks.reset_u) {
    memcpy(u, &e->hooks.reset_u[(size_t)ORC_RESET_U_STRIDE * i], sizeof u);
  } else {
    draw4(e->seed, gid, step_ctr, PURPOSE_RESET_A, &u[0]);
    draw4(e->seed, gid, step_ctr, PURPOSE_RESET_B, &u[4]);
  }
  float pos[3], q[4], cmd[4];
  orc_reset_draws(c, u, pos, q, cmd);
  if (level_mode(e)) {
    /* `if not self.curriculum:` skips the re-draw (KAMMA.py:490): commands[:, 0] keeps the draw of :448,
     * commands[:, 2] its previous value, both times the small-command mask of :458 */
    float x0 = c->cmd_x_span * u[3] + c->cmd_x_lo, y0 = c->cmd_y_span * u[4] + c->cmd_y_lo;
    float keep = sqrtf(x0 * x0 + y0 * y0) > 0.25f ? 1.0f : 0.0f;
    cmd[0] = x0 * keep;
    cmd[2] = e->commands[4 * i + 2] * keep;
  } else if (c->cmd_sampler == OIS_CMD_BOX) {
    /* KAMMA_curr1.py:577-598 */
    float xmin, xmax;
    orc_box_bounds(cnt[1], c->max_sim_steps, c->cmd_x_lo, c->cmd_x_hi, &xmin, &xmax);
    float xv = rand_float(xmin, xmax, u[6]);
    cmd[0] = xv > 0.0f ? xv : 0.0f;
  } else if (c->cmd_sampler == OIS_CMD_EVAL) {
    cmd[0] = orc_eval_command(cnt[1], c->eval_ramp_steps);
  } else if (c->cmd_sampler == OIS_CMD_GRID) {
    /* KAMMA_curr2.py:602-640; u[6] picks the side (torch.rand(1) < 0.5), a third Philox block feeds the normal
     * draw (the omega draw of :631-636 is discarded) */
    float uc[4], z;
    if (e->hooks.reset_u) {
      z = e->hooks.reset_u[(size_t)ORC_RESET_U_STRIDE * i + 8];
    } else {
      draw4(e->seed, gid, step_ctr, PURPOSE_RESET_C, uc);
      z = normal_det(uc[0], uc[1]);
    }
    cmd[0] = orc_grid_command(cnt[1], c->max_sim_steps, c->cmd_x_lo, c->cmd_x_hi, u[6], z);
  }
  /* On the arena's own plane (OIS_BUF_PARAMS: unit normal n, through the origin; build-defined, SPEC 3.2) the robot is set down ON
   * that plane: the base link at its height above the ground point under (x, y), along n, and the drawn orientation turned by the
   * shortest rotation that takes the world z axis to n. With n = (0, 0, 1) — the reference's ground plane — this is the identity
   * (up to the sign of a zero). Before round 4 the pose ignored the plane: on a slope of 0.3 a robot drawn 1 m uphill started
   * 0.3 m inside the ground and was thrown out tumbling. Not on a height field, whose tiles carry their own origins (SPEC 3.8). */
  if (!e->terrain_on) {
    const float* prm = &e->params[4 * i];
    const float nx = prm[1], ny = prm[2], nz = prm[3], h = pos[2];
    const float zg = -(fmaf(ny, pos[1], nx * pos[0]) / nz);
    const float a = 1.0f + nz, d = sqrtf(a + a);
    /* q <- q_n (x) q with q_n = (a, -n_y, n_x, 0) / d, as the plain Hamilton product (exact for the identity, which the
     * reference's sum-of-products form :434 is not) */
    const float w1 = a / d, x1 = -ny / d, y1 = nx / d;
    const float w2 = q[0], x2 = q[1], y2 = q[2], z2 = q[3];
    q[0] = fmaf(-y1, y2, fmaf(-x1, x2, w1 * w2));
    q[1] = fmaf(y1, z2, fmaf(x1, w2, w1 * x2));
    q[2] = fmaf(-x1, z2, fmaf(y1, w2, w1 * y2));
    q[3] = fmaf(x1, y2, fmaf(-y1, x2, w1 * z2));
    pos[0] = fmaf(h, nx, pos[0]);
    pos[1] = fmaf(h, ny, pos[1]);
    pos[2] = fmaf(h, nz, zg);
  }
  /* pose: the reference sets the base-link pose (:441-443); the state stores the COM */
  float* P = &e->pose[4 * i];
  float R[9];
  rotmat(q, R);
  P[0] = fmaf(R[2], c->com_height, pos[0]);
  P[1] = fmaf(R[5], c->com_height, pos[1]);
  P[2] = fmaf(R[8], c->com_height, pos[2]);
  memcpy(&e->quat[4 * i], q, sizeof q);
  for (int k = 0; k < 3; ++k) {
    e->linvel[4 * i + k] = c->init_lin_vel[k]; /* :435 */
    e->angvel[4 * i + k] = c->init_ang_vel[k];
  }
  for (int k = 0; k < 4; ++k) e->wheel_vel[4 * i + k] = c->init_dof_vel[k]; /* :438, :446 */
  for (int k = 0; k < 4; ++k) e->lam_n[4 * i + k] = e->lam_l[4 * i + k] = e->lam_t[4 * i + k] = 0; /* teleport: drop the contact cache */
  memcpy(&e->commands[4 * i], cmd, sizeof cmd);
  cnt[0] = 0;   /* :462 */
  cnt[2] += 1;  /* :463 */
}

/* pairwise tree over the GROUP arenas of a group; absent arenas contribute `ident` */
static float tree16(const float* x, int n, float ident, int op) {
  float t[GROUP];
  for (int i = 0; i < GROUP; ++i) t[i] = i < n ? x[i] : ident;
  for (int s = 1; s < GROUP; s <<= 1)
    for (int i = 0; i < GROUP; i += 2 * s) {
      float a = t[i], b = t[i + s];
      t[i] = op == 0 ? a + b : (op == 1 ? fminf(a, b) : fmaxf(a, b));
    }
  return t[0];
}

static void group_reduce(orc_env* e, int g, const uint8_t* rs, const reset_contrib* rc, int32_t stamp,
                         float gamma1, float gamma2, int count_stats) {
  int base = g * GROUP;
  int n = e->N - base < GROUP ? e->N - base : GROUP;
  int any = 0;
  for (int i = 0; i < n; ++i) any |= rs[base + i];
  if (!any) return;
  float* P = &e->partials[(size_t)g * PART_STRIDE];
  float x[GROUP];
  for (int k = 0; k < 16; ++k) {
    for (int i = 0; i < n; ++i) x[i] = rs[base + i] ? rc[base + i].v[k] : 0.0f;
    P[k] = tree16(x, n, 0.0f, 0);
  }
  P[P_G1] = gamma1; P[P_G2] = gamma2;
  for (int i = 0; i < n; ++i) x[i] = rs[base + i] ? rc[base + i].sm : 0.0f;
  P[P_SM] = tree16(x, n, 0.0f, 0);
  for (int i = 0; i < n; ++i) x[i] = rs[base + i] ? rc[base + i].te : 0.0f;
  P[P_TE] = tree16(x, n, 0.0f, 0);
  for (int i = 0; i < n; ++i) x[i] = rs[base + i] ? rc[base + i].dv : 0.0f;
  P[P_DV] = tree16(x, n, 0.0f, 0);
  for (int i = 0; i < n; ++i) x[i] = rs[base + i] ? rc[base + i].ret * rc[base + i].ret : 0.0f;
  P[P_RET2] = tree16(x, n, 0.0f, 0);
  for (int i = 0; i < n; ++i) x[i] = rs[base + i] ? rc[base + i].ret : INFINITY;
  P[P_RMIN] = tree16(x, n, INFINITY, 1);
  for (int i = 0; i < n; ++i) x[i] = rs[base + i] ? rc[base + i].ret : -INFINITY;
  P[P_RMAX] = tree16(x, n, -INFINITY, 2);
  for (int i = 0; i < n; ++i) x[i] = rs[base + i] ? rc[base + i].len : 0.0f;
  P[P_LEN] = tree16(x, n, 0.0f, 0);
  e->stamps[g] = stamp;
  if (count_stats) {
    double* A = &e->stat_acc[(size_t)g * ST_STRIDE];
    A[ST_COUNT] += (double)P[OIS_EI_COUNT];
    A[ST_RET] += (double)P[OIS_EI_FINAL];
    A[ST_RET2] += (double)P[P_RET2];
    A[ST_RMIN] = fmin(A[ST_RMIN], (double)P[P_RMIN]);
    A[ST_RMAX] = fmax(A[ST_RMAX], (double)P[P_RMAX]);
    A[ST_LEN] += (double)P[P_LEN];
  }
}

static void stats_clear(orc_env* e) {
  for (int g = 0; g < e->G; ++g) {
    double* A = &e->stat_acc[(size_t)g * ST_STRIDE];
    memset(A, 0, ST_STRIDE * sizeof(double));
    A[ST_RMIN] = INFINITY;
    A[ST_RMAX] = -INFINITY;
  }
  e->steps_done = 0.0;
}

/* post_reset (KAMMA.py:366-414): zero task buffers, reset_idx(all) */
int orc_reset(orc_env* e, uint64_t seed) {
  if (!e) return fail(OIS_EINVAL, "null env");
  int N = e->N;
  e->seed = seed;
  size_t f4 = (size_t)N * 4 * sizeof(float);
  memset(e->obs, 0, (size_t)N * 6 * sizeof(float)); memset(e->rew, 0, (size_t)N * sizeof(float));
  memset(e->wheel_torqs, 0, f4); memset(e->pose, 0, f4); memset(e->quat, 0, f4); memset(e->linvel, 0, f4);
  memset(e->angvel, 0, f4); memset(e->wheel_vel, 0, f4); memset(e->commands, 0, f4);
  memset(e->sums_a, 0, f4); memset(e->sums_b, 0, f4); memset(e->sums_c, 0, f4);
  memset(e->lam_n, 0, f4 / 2); memset(e->lam_l, 0, f4 / 2); memset(e->lam_t, 0, f4 / 2);
  memset(e->cmd_fraction, 0, (size_t)N * sizeof(float)); /* :498 at post_reset: the sums are zero */
  memset(e->done, 0, (size_t)N); memset(e->timeout, 0, (size_t)N);
  memset(e->counters, 0, (size_t)N * 4 * sizeof(int32_t));
  memset(e->partials, 0, (size_t)e->G * PART_STRIDE * sizeof(float));
  memset(e->episode_info, 0, sizeof e->episode_info);
  memset(e->env0_diag, 0, sizeof e->env0_diag);
  stats_clear(e);
  reset_contrib* rc = (reset_contrib*)zalloc((size_t)N * sizeof *rc);
  uint8_t* rs = (uint8_t*)malloc((size_t)N);
  if (!rc || !rs) { free(rc); free(rs); return fail(OIS_ENOMEM, "host alloc"); }
  memset(rs, 1, (size_t)N);
  /* gamma_assist1/2 are their constructor values here: ones (KAMMA.py:192-193), or the
   * configured constants for the constant-gamma variants */
  float g1 = e->cfg.gamma1_mode == OIS_GAMMA_CONST ? e->cfg.gamma1_const : 1.0f;
  float g2 = e->cfg.gamma2_mode == OIS_GAMMA_CONST ? e->cfg.gamma2_const : 1.0f;
  for (int i = 0; i < N; ++i) reset_arena(e, i, 0u, g2, &rc[i]);
  for (int g = 0; g < e->G; ++g) group_reduce(e, g, rs, rc, 0, g1, g2, 0);
  free(rc); free(rs);
  e->old_torques_zeroed = 1;
  e->reset_done = 1;
  return OIS_OK;
}

/* one policy step of arena i: pre_physics_step + substeps + post_physics_step.
 * Returns reward. rs/rc: reset flag and contribution (for the group reduction). */
/* KAMMA_eval.py:677-683 (pre_physics_step): Smoothness += sum((old_torques - wheel_torqs)^2) / 10, Desired velocity +=
 * desired_v / 10; :840 (calculate_metrics): Tracking error += |v_delta| / 10 (skipped when v_delta == NULL).
 * sums_c4 = [Smoothness, Tracking error, Desired velocity, -] per arena. The wheel sum in the SPEC order. */
void orc_eval_sums(int n, const float* old4, const float* wt4, const float* desired_v, const float* v_delta, float* sums_c4) {
  for (int i = 0; i < n; ++i) {
    const float* o = &old4[4 * i];
    const float* w = &wt4[4 * i];
    float* sc = &sums_c4[4 * i];
    float d0 = o[0] - w[0], d1 = o[1] - w[1], d2 = o[2] - w[2], d3 = o[3] - w[3];
    sc[0] += ((d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3)) / 10.0f;
    sc[2] += desired_v[i] / 10.0f;
    if (v_delta) sc[1] += fabsf(v_delta[i]) / 10.0f;
  }
}

static float step_arena(orc_env* e, int i, const float* act /*[A]*/, uint8_t* rs, reset_contrib* rc, float* g1_out, float* g2_out) {
  const ois_config* c = &e->cfg;
  const derived* D = &e->D;
  float* P = &e->pose[4 * i];
  float* Q = &e->quat[4 * i];
  float* V = &e->linvel[4 * i];
  float* W = &e->angvel[4 * i];
  float* JV = &e->wheel_vel[4 * i];
  float* CMD = &e->commands[4 * i];
  int32_t* cnt = &e->counters[4 * i];
  float* sa = &e->sums_a[4 * i];
  float* sb = &e->sums_b[4 * i];
  float* sc = &e->sums_c[4 * i];
  const float* prm = &e->params[4 * i];
  int64_t gid = c->arena_offset + i;

  /* ---- pre_physics_step (KAMMA.py:596-656) ---- */
  float a4[4];
  if (c->action_dim == 4) { a4[0] = act[0]; a4[1] = act[1]; a4[2] = act[2]; a4[3] = act[3]; }
  else if (c->action_dim == 2) { a4[0] = a4[1] = act[0]; a4[2] = a4[3] = act[1]; } /* KA_DDPG_2D.py:623-624 */
  else { a4[0] = a4[1] = a4[2] = a4[3] = act[0]; }                                  /* KA_DDPG_1D.py:600-603 */
  float desired_v = CMD[0], desired_omega = CMD[2];
  float current_v = P[3];      /* v_forward_projected of the previous step :607 */
  float current_omega = W[2];  /* base_velocities[:,5] :608 */
  float v_delta, omega_delta, ac_left, ac_right;
  criteria1(desired_v, desired_omega, current_v, current_omega, c->Kp, c->Kp_omega, c->vehicle_mass,
            c->vehicle_inertia, c->policy_dt, &v_delta, &omega_delta, &ac_left, &ac_right);
  float gsched = gamma1f(cnt[1], c->max_sim_steps);
  float gamma1 = c->gamma1_mode == OIS_GAMMA_CONST ? c->gamma1_const : gsched;
  float gamma2 = c->gamma2_mode == OIS_GAMMA_CONST ? c->gamma2_const : gsched;
  float rnd = 0.0f;
  if (c->mix_mode == OIS_MIX_SWITCH) {
    float u[4];
    draw4(e->seed, gid, (uint32_t)cnt[1], PURPOSE_MASK, u); /* torch.rand(num_envs,1) :633 */
    rnd = e->hooks.switch_u ? e->hooks.switch_u[i] : u[0];
  }
  float torques[4], wt[4], guiding;
  mix1(c->mix_mode, c->guiding_mode, a4, ac_left, ac_right, gamma1, rnd, c->action_scale, c->torque_clip,
       torques, wt, &guiding);
  if (c->eval_sums) {
    /* KAMMA_eval.py:677-685. old_torques = the previous step's clipped torques, which still sit in the
     * wheel_torqs buffer — or zeros when a reset_idx call came in between: it zeroes old_torques of ALL
     * arenas (KAMMA_eval.py:470), whichever arena reset (sums_c[3] == 0: before this arena's first step) */
    const float* old = &e->wheel_torqs[4 * i];
    float o4[4];
    const int zeroed = sc[3] == 0.0f || e->old_torques_zeroed;
    for (int k = 0; k < 4; ++k) o4[k] = zeroed ? 0.0f : old[k];
    orc_eval_sums(1, o4, wt, &desired_v, NULL, sc);
    sc[3] = 1.0f; /* this arena has stepped since its reset */
  }
  for (int k = 0; k < 4; ++k) e->wheel_torqs[4 * i + k] = wt[k];

  /* ---- physics: decimation substeps (:646-653) + controlFrequencyInv substeps (:43) ---- */
  float p[3] = {P[0], P[1], P[2]}, q[4] = {Q[0], Q[1], Q[2], Q[3]};
  float v[3] = {V[0], V[1], V[2]}, w[3] = {W[0], W[1], W[2]};
  float n[3] = {prm[1], prm[2], prm[3]};
  float mu = prm[0];
  float R[9], wb[3], spin[4];
  rotmat(q, R);
  rotT(R, w, wb);
  for (int k = 0; k < 4; ++k) spin[k] = JV[k] + wb[1];
  int nsub = c->decimation + c->extra_substeps;
  float LN[4], LL[4], LT[4]; /* the stored binary16 impulses, widened */
  for (int k = 0; k < 4; ++k) { LN[k] = orc_h2f(e->lam_n[4 * i + k]); LL[k] = orc_h2f(e->lam_l[4 * i + k]); LT[k] = orc_h2f(e->lam_t[4 * i + k]); }
  rcache C;
  const terrain_ctx* T = e->terrain_on ? &e->terrain : NULL;
  terrain_carry TC;
  for (int s = 0; s < nsub; ++s) substep(D, mu, n, wt, p, q, v, wb, spin, LN, LL, LT, &C, s, T, &e->env_origins[4 * i], &TC);
  /* Safety net (build-defined): a Newton refresh diverges if a row's effective mass more than
   * doubles between two substeps (needs > ~50 rad/s of tumbling) and could leave a non-finite
   * state; such an arena would never reset (every comparison of is_done is false on NaN) and
   * would poison the policy's batch. A non-finite outcome is discarded: the arena keeps its
   * pre-step pose at rest and is reset as fallen. */
  /* the impulses cross a policy-step boundary in binary16 (SPEC 3.5); rounded before the check, so an
   * impulse beyond the half range (inf) counts as non-finite */
  for (int k = 0; k < 4; ++k) { LN[k] = round_half(LN[k]); LL[k] = round_half(LL[k]); LT[k] = round_half(LT[k]); }
  float chk = 0.0f;
  for (int k = 0; k < 3; ++k) chk += fabsf(p[k]) + fabsf(v[k]) + fabsf(wb[k]);
  for (int k = 0; k < 4; ++k) chk += fabsf(q[k]) + fabsf(spin[k]) + fabsf(LN[k]) + fabsf(LL[k]) + fabsf(LT[k]);
  /* ... and a linear speed no robot of this task reaches (SPEC 3.9: 50 m/s; the fastest legitimate one measured is a fall off the
   * field's relief, 8 m/s): a robot on its side has n_b along the axle, its contact points jump by a wheel radius within a substep
   * and the Newton-refreshed gains (SPEC 3.6) are then finite garbage — 180 to 4 000 m/s launches, two in 3 x 10^6 arena-steps of
   * robots shoved over on the plane (round 4). Finite, so the sum above does not see them. */
  const float v2 = fmaf(v[2], v[2], fmaf(v[1], v[1], v[0] * v[0]));
  const int blown = !(chk < 3.0e38f) || !(v2 < 2500.0f);
  if (blown) {
    for (int k = 0; k < 3; ++k) { p[k] = P[k]; v[k] = 0.0f; wb[k] = 0.0f; }
    for (int k = 0; k < 4; ++k) { q[k] = Q[k]; spin[k] = 0.0f; LN[k] = LL[k] = LT[k] = 0.0f; }
  }
  rotmat(q, R);
  rot(R, wb, w); /* body -> world with the final orientation */
  if (blown) w[0] = w[1] = w[2] = 0.0f; /* R may itself be non-finite if the pose was corrupted from outside */
  float jv[4];
  for (int k = 0; k < 4; ++k) jv[k] = spin[k] - wb[1];
  if (e->hooks.physics_in) { /* fixture replay: the simulator's outcome comes from the fixture */
    const float* s = &e->hooks.physics_in[(size_t)ORC_PHYSICS_STRIDE * i];
    for (int k = 0; k < 3; ++k) { p[k] = s[k]; v[k] = s[7 + k]; w[k] = s[10 + k]; }
    for (int k = 0; k < 4; ++k) { q[k] = s[3 + k]; jv[k] = s[13 + k]; }
    rotmat(q, R);
  }
  if (e->hooks.physics_out) {
    float* s = &e->hooks.physics_out[(size_t)ORC_PHYSICS_STRIDE * i];
    for (int k = 0; k < 3; ++k) { s[k] = p[k]; s[7 + k] = v[k]; s[10 + k] = w[k]; }
    for (int k = 0; k < 4; ++k) { s[3 + k] = q[k]; s[13 + k] = jv[k]; }
    s[17] = fmaf(-R[2], c->com_height, p[0]); s[18] = fmaf(-R[5], c->com_height, p[1]); s[19] = fmaf(-R[8], c->com_height, p[2]);
  }

  /* ---- post_physics_step (:674-727) ---- */
  cnt[0] += 1; /* episode_buf :675 */
  cnt[1] += 1; /* sim_steps   :676 */
  /* base-link position for the bounds check */
  float plink[3] = {fmaf(-R[2], c->com_height, p[0]), fmaf(-R[5], c->com_height, p[1]), fmaf(-R[8], c->com_height, p[2])};
  uint8_t reset, timeout;
  is_done1(cnt[0], c->max_episode_length, q, plink, c->bounds, &reset, &timeout); /* :687 */
  if (blown) reset = 1;
  float S[8] = {sa[0], sa[1], sa[2], sa[3], sb[0], sb[1], sb[2], sb[3]};
  float prev_lin = V[3], prev_ang = W[3];
  metrics_out mo;
  metrics1(current_v, w[2], &prev_lin, &prev_ang, v_delta, omega_delta, wt, guiding, gamma2, reset, timeout,
           c->policy_dt, c->w1, c->w2, c->w3, c->sparse_bonus, c->sparse_threshold, c->terminal_reward, S, &mo); /* :689 */
  for (int k = 0; k < 4; ++k) { sa[k] = S[k]; sb[k] = S[4 + k]; }
  if (c->eval_sums) sc[1] += fabsf(v_delta) / 10.0f; /* KAMMA_eval.py:840 */
  /* write back the post-physics state (reset_arena overwrites it if the arena resets) */
  P[0] = p[0]; P[1] = p[1]; P[2] = p[2];
  Q[0] = q[0]; Q[1] = q[1]; Q[2] = q[2]; Q[3] = q[3];
  V[0] = v[0]; V[1] = v[1]; V[2] = v[2]; V[3] = prev_lin;
  W[0] = w[0]; W[1] = w[1]; W[2] = w[2]; W[3] = prev_ang;
  for (int k = 0; k < 4; ++k) JV[k] = jv[k];
  for (int k = 0; k < 4; ++k) { e->lam_n[4 * i + k] = orc_f2h(LN[k]); e->lam_l[4 * i + k] = orc_f2h(LL[k]); e->lam_t[4 * i + k] = orc_f2h(LT[k]); }
  rs[i] = reset;
  if (reset) reset_arena(e, i, (uint32_t)cnt[1], gamma2, &rc[i]); /* :691-693 */
  /* get_observations (:825-833): desired_* are views of commands -> post-reset values */
  float o[6] = {CMD[0], CMD[2], v_delta, omega_delta, mo.lin_acc, mo.ang_acc};
  float ob[6];
  for (int k = 0; k < 6; ++k) ob[k] = clipf(o[k], -c->clip_obs, c->clip_obs); /* rl_task.py:99 clamp in the VecEnv */
  /* KAMMA_g1d_g21.py:838-842: get_observations replaces a (nearly) zero rew_buf[0] by 1e-12 before it is
   * returned (division guard of the env-0 percentages); the episode sums were added in calculate_metrics */
  if (c->env0_reward_guard && gid == 0 && fabs((double)mo.rew) < 1e-12) mo.rew = 1e-12f;
  if (e->rp_size > 0) {
    /* FIFOMemory row (train_kaddpg_4d_nofifo_random_RLIL.py:147-156): states = the previous
     * observation, actions = executed torques / action_scale (base_KA_DDPG.py:212) */
    size_t row = (size_t)e->rp_head * e->N + i;
    for (int k = 0; k < 6; ++k) { e->rp_states[row * 6 + k] = e->obs[6 * i + k]; e->rp_next[row * 6 + k] = ob[k]; }
    for (int k = 0; k < 4; ++k) e->rp_actions[row * 4 + k] = wt[k] / c->action_scale;
    e->rp_rewards[row] = mo.rew;
    e->rp_done[row] = reset;
  }
  for (int k = 0; k < 6; ++k) e->obs[6 * i + k] = ob[k];
  e->rew[i] = mo.rew;
  e->done[i] = reset;
  e->timeout[i] = timeout;
  /* tail of post_physics_step (:701-710) on the (possibly reset) state */
  P[3] = forward_projection1(Q, V);
  if (i == 0) {
    float* d = e->env0_diag; /* KAMMA.py:840-867 ingredients */
    d[0] = CMD[0]; d[1] = current_v; d[2] = CMD[2]; d[3] = current_omega; d[4] = v_delta; d[5] = omega_delta;
    d[6] = mo.lin_acc; d[7] = mo.ang_acc; d[8] = (float)cnt[2];
    for (int k = 0; k < 4; ++k) { d[9 + k] = torques[k]; d[15 + k] = c->action_scale * a4[k]; }
    d[13] = ac_left; d[14] = ac_right;
    d[19] = gamma2; d[20] = mo.comp1; d[21] = mo.comp2; d[22] = mo.comp3; d[23] = mo.rdense; d[24] = mo.rsparse;
    d[25] = mo.robs; d[26] = guiding; d[28] = gamma1;
    /* the divisor of the env-0 percentages: guarded for the arena that logs = local arena 0 of every environment, shards too */
    d[27] = (c->env0_reward_guard && fabs((double)mo.rew) < 1e-12) ? 1e-12f : mo.rew;
  }
  *g1_out = gamma1; *g2_out = gamma2;
  return mo.rew;
}

static int step_all(orc_env* e, int policy, const float* actions, float* reward_sum, reset_contrib* rc, uint8_t* rs) {
  const int N = e->N, A = e->cfg.action_dim, G = e->G;
  int32_t stamp = e->counters[1] + 1; /* sim_steps after this step (identical for all arenas) */
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(e->threads)
#endif
  for (int g = 0; g < G; ++g) {
    int base = g * GROUP;
    int n = N - base < GROUP ? N - base : GROUP;
    float lg1 = 0.0f, lg2 = 0.0f;
    for (int j = 0; j < n; ++j) {
      int i = base + j;
      float a[4] = {0, 0, 0, 0};
      if (policy == OIS_POLICY_EXTERNAL) {
        for (int k = 0; k < A; ++k) a[k] = actions[(size_t)i * A + k];
      } else if (policy == OIS_POLICY_RANDOM) {
        float u[4];
        draw4(e->seed, e->cfg.arena_offset + i, (uint32_t)e->counters[4 * i + 1], PURPOSE_ACTION, u);
        for (int k = 0; k < A; ++k) a[k] = u[k];
      }
      float r = step_arena(e, i, a, rs, rc, &lg1, &lg2);
      if (reward_sum) reward_sum[i] += r;
    }
    group_reduce(e, g, rs, rc, stamp, lg1, lg2, 1);
  }
  if (level_mode(e)) {
    /* KAMMA.py:498: any reset_idx call re-snapshots `fraction` for ALL arenas (after the sums of the
     * resetting ones were zeroed); then the tail of post_physics_step re-draws the command of every
     * arena (KAMMA.py:719-725). episode_step_counter == sim_steps: both count policy steps since post_reset. */
    int any = 0;
    for (int i = 0; i < N; ++i) any |= rs[i];
    if (any) for (int i = 0; i < N; ++i) e->cmd_fraction[i] = e->sums_a[4 * i] / e->cfg.curriculum_threshold_high;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(e->threads)
#endif
    for (int i = 0; i < N; ++i) {
      float u[4];
      draw4(e->seed, e->cfg.arena_offset + i, (uint32_t)stamp, PURPOSE_CMD, u);
      int level = e->terrain_on ? e->terrain_idx[4 * i] : 0; /* zeros without a terrain (KAMMA.py:223) */
      if (e->hooks.levels) level = e->hooks.levels[i];
      const float z = e->hooks.level_z ? e->hooks.level_z[i] : normal_det(u[0], u[1]);
      orc_level_command(level, e->cmd_fraction[i], stamp, z, (double)e->cfg.decimation * (double)e->cfg.sim_dt,
                        (double)e->cfg.episode_length_s, &e->commands[4 * i], &e->commands[4 * i + 2]);
    }
  }
  if (e->cfg.eval_sums) { /* KAMMA_eval.py:470: a reset_idx call zeroes old_torques of all arenas */
    int any = 0;
    for (int i = 0; i < N; ++i) any |= rs[i];
    e->old_torques_zeroed = any;
  }
  e->steps_done += (double)N;
  if (e->rp_size > 0) {
    if (e->rp_head + 1 >= e->rp_size) e->rp_filled = 1;
    e->rp_head = (e->rp_head + 1) % e->rp_size;
  }
  return OIS_OK;
}

int orc_step(orc_env* e, const float* actions) {
  if (!e || !actions) return fail(OIS_EINVAL, "null argument");
  if (!e->reset_done) return fail(OIS_ESTATE, "orc_step before orc_reset");
  reset_contrib* rc = (reset_contrib*)malloc((size_t)e->N * sizeof *rc);
  uint8_t* rs = (uint8_t*)calloc((size_t)e->N, 1);
  if (!rc || !rs) { free(rc); free(rs); return fail(OIS_ENOMEM, "host alloc"); }
  int r = step_all(e, OIS_POLICY_EXTERNAL, actions, NULL, rc, rs);
  free(rc); free(rs);
  return r;
}

int orc_rollout(orc_env* e, int k, int policy, const float* actions, float* reward_sum) {
  if (!e || k < 1) return fail(OIS_EINVAL, "bad argument");
  if (policy == OIS_POLICY_EXTERNAL && !actions) return fail(OIS_EINVAL, "EXTERNAL policy needs actions");
  if (policy < 0 || policy > 2) return fail(OIS_EINVAL, "unknown policy");
  if (!e->reset_done) return fail(OIS_ESTATE, "orc_rollout before orc_reset");
  reset_contrib* rc = (reset_contrib*)malloc((size_t)e->N * sizeof *rc);
  uint8_t* rs = (uint8_t*)calloc((size_t)e->N, 1);
  if (!rc || !rs) { free(rc); free(rs); return fail(OIS_ENOMEM, "host alloc"); }
  if (reward_sum) memset(reward_sum, 0, (size_t)e->N * sizeof(float));
  for (int s = 0; s < k; ++s) {
    const float* a = policy == OIS_POLICY_EXTERNAL ? actions + (size_t)s * e->N * e->cfg.action_dim : NULL;
    memset(rs, 0, (size_t)e->N);
    step_all(e, policy, a, reward_sum, rc, rs);
  }
  free(rc); free(rs);
  return OIS_OK;
}

/* FIFO transition memory, mirror of ois_replay_attach / _buffer / _cursor */
int orc_replay_attach(orc_env* e, int memory_size) {
  if (!e || memory_size < 0) return fail(OIS_EINVAL, "bad argument");
  free(e->rp_states); free(e->rp_actions); free(e->rp_rewards); free(e->rp_next); free(e->rp_done);
  e->rp_states = e->rp_actions = e->rp_rewards = e->rp_next = NULL; e->rp_done = NULL;
  e->rp_size = memory_size; e->rp_head = 0; e->rp_filled = 0;
  if (memory_size > 0) {
    size_t rows = (size_t)memory_size * (size_t)e->N;
    e->rp_states = zalloc(rows * 6 * sizeof(float)); e->rp_actions = zalloc(rows * 4 * sizeof(float));
    e->rp_rewards = zalloc(rows * sizeof(float)); e->rp_next = zalloc(rows * 6 * sizeof(float));
    e->rp_done = zalloc(rows);
    if (!e->rp_states || !e->rp_actions || !e->rp_rewards || !e->rp_next || !e->rp_done) return fail(OIS_ENOMEM, "host alloc");
  }
  return OIS_OK;
}
int orc_replay_buffer(orc_env* e, int id, void** ptr, size_t* bytes) {
  if (!e || !ptr || !bytes || e->rp_size <= 0) return fail(OIS_EINVAL, "no replay memory");
  size_t rows = (size_t)e->rp_size * (size_t)e->N;
  switch (id) {
    case OIS_RP_STATES: *ptr = e->rp_states; *bytes = rows * 6 * sizeof(float); break;
    case OIS_RP_ACTIONS: *ptr = e->rp_actions; *bytes = rows * 4 * sizeof(float); break;
    case OIS_RP_REWARDS: *ptr = e->rp_rewards; *bytes = rows * sizeof(float); break;
    case OIS_RP_NEXT_STATES: *ptr = e->rp_next; *bytes = rows * 6 * sizeof(float); break;
    case OIS_RP_TERMINATED: *ptr = e->rp_done; *bytes = rows; break;
    default: return fail(OIS_EINVAL, "unknown replay buffer id");
  }
  return OIS_OK;
}
int orc_replay_cursor(const orc_env* e, int* memory_index, int* filled) {
  if (!e || !memory_index || !filled) return fail(OIS_EINVAL, "null argument");
  *memory_index = e->rp_head; *filled = e->rp_filled;
  return OIS_OK;
}

/* ---- height-field terrain: attach + get_heights ---- */
int orc_terrain_attach(orc_env* e, const ois_terrain* t, const int16_t* heights, const float* terrain_origins,
                       const int32_t* levels_types) {
  if (!e) return fail(OIS_EINVAL, "null env");
  free(e->theights); free(e->terrain_origins);
  e->theights = NULL; e->terrain_origins = NULL; e->terrain_on = 0;
  memset(e->env_origins, 0, (size_t)e->N * 4 * sizeof(float));
  memset(e->terrain_idx, 0, (size_t)e->N * 4 * sizeof(int32_t));
  if (!t) return OIS_OK;
  if (t->curriculum && e->cfg.cmd_sampler == OIS_CMD_EVAL) /* KAMMA_eval.py:544: evalsampling is tested before curriculum */
    return fail(OIS_EINVAL, "terrain.curriculum with the EvalSampling commands is not supported");
  if (t->struct_bytes != sizeof(ois_terrain)) return fail(OIS_EINVAL, "ois_terrain.struct_bytes mismatch");
  if (t->rows < 2 || t->cols < 2 || !(t->horizontal_scale > 0.0f) || t->num_levels < 1 || t->num_types < 1 ||
      t->max_level < 0 || t->max_level >= t->num_levels || !heights || !terrain_origins || !levels_types)
    return fail(OIS_EINVAL, "bad terrain description");
  if (t->footprint_stride < 0 || t->footprint_stride > 2) return fail(OIS_EINVAL, "ois_terrain.footprint_stride must be 1 or 2 (0 = 1)");
  for (int i = 0; i < e->N; ++i)
    if (levels_types[2 * i] < 0 || levels_types[2 * i] >= t->num_levels || levels_types[2 * i + 1] < 0 ||
        levels_types[2 * i + 1] >= t->num_types)
      return fail(OIS_EINVAL, "terrain level / type out of range");
  size_t cells = (size_t)t->rows * (size_t)t->cols, no = (size_t)t->num_levels * t->num_types * 3;
  e->theights = (int16_t*)malloc(cells * sizeof(int16_t));
  e->terrain_origins = (float*)malloc(no * sizeof(float));
  if (!e->theights || !e->terrain_origins) return fail(OIS_ENOMEM, "host alloc");
  memcpy(e->theights, heights, cells * sizeof(int16_t));
  memcpy(e->terrain_origins, terrain_origins, no * sizeof(float));
  e->tcfg = *t;
  e->terrain.h = e->theights; e->terrain.rows = t->rows; e->terrain.cols = t->cols;
  e->terrain.hs = t->horizontal_scale; e->terrain.inv_hs = 1.0f / t->horizontal_scale;
  e->terrain.vs = t->vertical_scale; e->terrain.border = t->border_size;
  e->terrain.stride = t->footprint_stride == 2 ? 2 : 1;
  for (int i = 0; i < e->N; ++i) {
    int lvl = levels_types[2 * i], typ = levels_types[2 * i + 1];
    e->terrain_idx[4 * i] = lvl; e->terrain_idx[4 * i + 1] = typ;
    const float* o = &e->terrain_origins[((size_t)lvl * t->num_types + typ) * 3];
    e->env_origins[4 * i] = o[0]; e->env_origins[4 * i + 1] = o[1]; e->env_origins[4 * i + 2] = o[2]; /* KAMMA.py:397 */
  }
  e->terrain_on = 1;
  return OIS_OK;
}

/* get_heights (KAMMA.py:872-893) for n robots given WORLD base positions and orientations.
 * Grid KAMMA.py:278-292 (14 x 10 points, meshgrid 'ij'); quat_apply_yaw KAMMA.py:895-899 with
 * normalize() = x / max(|x|, 1e-9) and quat_apply(a, b) = b + w t + xyz x t, t = 2 xyz x b
 * (omni.isaac.core.utils.torch, third-party, published formulas). out: [n, 140] metres. */
void orc_get_heights(int n, const float* pos3, const float* quat4, const int16_t* heights, int rows, int cols,
                     float horizontal_scale, float vertical_scale, float border_size, float* out) {
  static const int GX[14] = {-8, -7, -6, -5, -4, -3, -2, 2, 3, 4, 5, 6, 7, 8};
  static const int GY[10] = {-5, -4, -3, -2, -1, 1, 2, 3, 4, 5};
  for (int i = 0; i < n; ++i) {
    float w = quat4[4 * i], z = quat4[4 * i + 3];
    float nrm = sqrtf(w * w + z * z); /* x, y are zeroed first (:897) */
    nrm = fmaxf(nrm, 1e-9f);
    float qw = w / nrm, qz = z / nrm;
    for (int a = 0; a < 14; ++a)
      for (int b = 0; b < 10; ++b) {
        float bx = 0.1f * (float)GX[a], by = 0.1f * (float)GY[b];
        float tx = -(qz * by) * 2.0f, ty = (qz * bx) * 2.0f; /* t = cross(xyz, b) * 2, xyz = (0, 0, qz) */
        float cx = -(qz * ty), cy = qz * tx;                 /* cross(xyz, t) */
        float X = (bx + qw * tx) + cx, Y = (by + qw * ty) + cy;
        X += pos3[3 * i]; Y += pos3[3 * i + 1];             /* + base_pos :878 */
        X += border_size; Y += border_size;                  /* :882 */
        float fx = X / horizontal_scale, fy = Y / horizontal_scale; /* :883, .long() truncates */
        fx = fminf(fmaxf(fx, 0.0f), (float)(rows - 2));      /* :886-887 clip after truncation == clamp before */
        fy = fminf(fmaxf(fy, 0.0f), (float)(cols - 2));
        int px = (int)fx, py = (int)fy;
        int16_t h1 = heights[(size_t)px * cols + py], h2 = heights[(size_t)(px + 1) * cols + (py + 1)]; /* :889-890 */
        out[(size_t)i * OIS_HEIGHT_POINTS + a * 10 + b] = (float)(h1 < h2 ? h1 : h2) * vertical_scale; /* :891-893 */
      }
  }
}

/* measured_heights of the current state: base-link position in the world = arena-local + env_origin */
int orc_height_scan(orc_env* e, float* out) {
  if (!e || !out) return fail(OIS_EINVAL, "null argument");
  if (!e->terrain_on) return fail(OIS_ESTATE, "no terrain attached");
  const ois_config* c = &e->cfg;
  for (int i = 0; i < e->N; ++i) {
    const float* P = &e->pose[4 * i];
    const float* Q = &e->quat[4 * i];
    const float* O = &e->env_origins[4 * i];
    float R[9];
    rotmat(Q, R);
    float pw[3] = {fmaf(-R[2], c->com_height, P[0]) + O[0], fmaf(-R[5], c->com_height, P[1]) + O[1], 0.0f};
    orc_get_heights(1, pw, Q, e->theights, e->terrain.rows, e->terrain.cols, e->terrain.hs, e->terrain.vs,
                    e->terrain.border, &out[(size_t)i * OIS_HEIGHT_POINTS]);
  }
  return OIS_OK;
}

/* extras["episode"] (KAMMA.py:480-488) from the most recent reset event */
int orc_episode_info(orc_env* e) {
  if (!e) return fail(OIS_EINVAL, "null env");
  /* two-level order (docs/SPEC.md §2.10): chunks of CHUNK_GROUPS consecutive groups are summed in
   * group order over the records carrying the chunk's newest stamp; the chunk partials whose
   * newest stamp is the global newest are summed in chunk order */
  int32_t smax = -1;
  for (int g = 0; g < e->G; ++g) smax = e->stamps[g] > smax ? e->stamps[g] : smax;
  float acc[PART_STRIDE];
  memset(acc, 0, sizeof acc);
  int first = 1;
  for (int g0 = 0; g0 < e->G; g0 += CHUNK_GROUPS) {
    int g1 = g0 + CHUNK_GROUPS < e->G ? g0 + CHUNK_GROUPS : e->G;
    int32_t cm = -1;
    for (int g = g0; g < g1; ++g) cm = e->stamps[g] > cm ? e->stamps[g] : cm;
    if (cm != smax) continue;
    float part[PART_STRIDE];
    memset(part, 0, sizeof part);
    for (int g = g0; g < g1; ++g) {
      if (e->stamps[g] != cm) continue;
      const float* P = &e->partials[(size_t)g * PART_STRIDE];
      for (int k = 0; k < 16; ++k) part[k] += P[k];
      part[P_SM] += P[P_SM]; part[P_TE] += P[P_TE]; part[P_DV] += P[P_DV];
      if (first) { acc[P_G1] = P[P_G1]; acc[P_G2] = P[P_G2]; first = 0; }
    }
    for (int k = 0; k < 16; ++k) acc[k] += part[k];
    acc[P_SM] += part[P_SM]; acc[P_TE] += part[P_TE]; acc[P_DV] += part[P_DV];
  }
  float* I = e->episode_info;
  memset(I, 0, sizeof e->episode_info);
  float cntf = acc[OIS_EI_COUNT];
  I[OIS_EI_COUNT] = cntf;
  float L = e->cfg.episode_length_s;
  for (int k = OIS_EI_R1; k <= OIS_EI_OBSERVED_OVER_FINAL; ++k) I[k] = (acc[k] / cntf) / L; /* :482-484 */
  I[OIS_EI_GAMMA1] = acc[P_G1];   /* :486 mean of N identical values */
  I[OIS_EI_GAMMA2] = acc[P_G2];   /* :487 */
  /* :488 torch.mean(terrain_levels.float()) over ALL arenas; zeros without a terrain (:223) */
  int64_t lsum = 0;
  if (e->hooks.levels) for (int i = 0; i < e->N; ++i) lsum += e->hooks.levels[i];
  else if (e->terrain_on) for (int i = 0; i < e->N; ++i) lsum += e->terrain_idx[4 * i];
  I[OIS_EI_TERRAIN_LEVEL] = (float)lsum / (float)e->N;
  I[OIS_EI_STEP] = (float)smax;
  I[OIS_EI_SMOOTHNESS] = (acc[P_SM] / cntf) / L;
  I[OIS_EI_TRACKING_ERROR] = (acc[P_TE] / cntf) / L;
  I[OIS_EI_DESIRED_VELOCITY] = (acc[P_DV] / cntf) / L;
  return OIS_OK;
}

int orc_stats_fetch(orc_env* e, ois_stats* out, int reset_after) {
  if (!e || !out) return fail(OIS_EINVAL, "null argument");
  double cnt = 0, ret = 0, ret2 = 0, len = 0, mn = INFINITY, mx = -INFINITY;
  for (int g0 = 0; g0 < e->G; g0 += CHUNK_GROUPS) { /* same two-level order as the extras */
    int g1 = g0 + CHUNK_GROUPS < e->G ? g0 + CHUNK_GROUPS : e->G;
    double c_cnt = 0, c_ret = 0, c_ret2 = 0, c_len = 0;
    for (int g = g0; g < g1; ++g) {
      const double* A = &e->stat_acc[(size_t)g * ST_STRIDE];
      c_cnt += A[ST_COUNT]; c_ret += A[ST_RET]; c_ret2 += A[ST_RET2]; c_len += A[ST_LEN];
      mn = fmin(mn, A[ST_RMIN]); mx = fmax(mx, A[ST_RMAX]);
    }
    cnt += c_cnt; ret += c_ret; ret2 += c_ret2; len += c_len;
  }
  out->episodes = cnt; out->return_sum = ret; out->return_sqsum = ret2; out->return_min = mn;
  out->return_max = mx; out->length_sum = len; out->steps = e->steps_done; out->reserved = 0.0;
  if (reset_after) stats_clear(e);
  return OIS_OK;
}

/* push_robots (KAMMA.py:729-733): base_velocities[:, 0:2] = torch_rand_float(-1.0, 1.0, (num_envs, 2)) */
int orc_push_robots(orc_env* e) {
  if (!e) return fail(OIS_EINVAL, "null env");
  for (int i = 0; i < e->N; ++i) {
    float u[4];
    draw4(e->seed, e->cfg.arena_offset + i, (uint32_t)e->counters[4 * i + 1], PURPOSE_PUSH, u);
    e->linvel[4 * i + 0] = rand_float(-1.0f, 1.0f, u[0]);
    e->linvel[4 * i + 1] = rand_float(-1.0f, 1.0f, u[1]);
  }
  return OIS_OK;
}

/* per-arena domain randomisation (BASELINE.json config 4) */
int orc_randomize_params(orc_env* e, float mu_lo, float mu_hi, const float* slopes, int n_slopes) {
  if (!e) return fail(OIS_EINVAL, "null env");
  if (n_slopes < 0 || (n_slopes > 0 && !slopes)) return fail(OIS_EINVAL, "bad slopes");
  if (!(mu_lo >= 0.0f) || !(mu_hi >= mu_lo) || !(mu_hi < 1e6f)) return fail(OIS_EINVAL, "friction range: 0 <= mu_lo <= mu_hi (finite)");
  for (int k = 0; k < n_slopes; ++k)
    if (!(fabsf(slopes[k]) <= 10.0f)) return fail(OIS_EINVAL, "slopes: |tan(incline)| <= 10, finite");
  for (int i = 0; i < e->N; ++i) {
    float u[4];
    draw4(e->cfg.seed, e->cfg.arena_offset + i, 0u, PURPOSE_PARAMS, u);
    float* prm = &e->params[4 * i];
    prm[0] = rand_float(mu_lo, mu_hi, u[0]);
    float nx = 0.0f, ny = 0.0f, nz = 1.0f;
    if (n_slopes > 0) {
      int idx = (int)(u[1] * (float)n_slopes);
      if (idx >= n_slopes) idx = n_slopes - 1;
      float sl = slopes[idx];
      float s, c;
      orc_sincos(3.14159274101257324f * u[2], &s, &c); /* heading in [0, pi); sign of the slope covers the rest */
      float inv = 1.0f / sqrtf(fmaf(sl, sl, 1.0f));
      nx = -(sl * c) * inv; ny = -(sl * s) * inv; nz = inv;
    }
    prm[1] = nx; prm[2] = ny; prm[3] = nz;
  }
  return OIS_OK;
}

/* ---- generalized advantage estimation (mirror of ois_gae) ----
 * mode OIS_GAE_PPO:  src/my_agents/ppo.py:345-358 (compute_gae incl. the normalisation);
 * mode OIS_GAE_PPOC: src/my_agents/ppoc.py:325-336 (compute_advantages_and_returns).
 * Tensors are row-major [T, N]; the loops run in the reference's order (t descending, one
 * rounding per torch op). Pinned by tests/golden/gae.npz (a torch loop transcription). */
int orc_gae(int mode, int T, int N, const float* rew, const uint8_t* term, const float* val,
            const float* last, double gamma, double lambda, int normalize, float* ret, float* adv) {
  if ((mode != OIS_GAE_PPO && mode != OIS_GAE_PPOC) || T <= 0 || N <= 0) return fail(OIS_EINVAL, "bad gae argument");
  if (!rew || !term || !val || !last || !ret || !adv) return fail(OIS_EINVAL, "null gae tensor");
  const float g = (float)gamma;
  const float lam = (float)lambda;           /* ppo: f32(lambda_coefficient) */
  const float gl = (float)(gamma * lambda);  /* ppoc: the Python double product meets the tensor */
  double s1 = 0.0, s2 = 0.0;
  for (int i = 0; i < N; ++i) {
    float nv = last[i];
    float a = 0.0f;
    for (int t = T - 1; t >= 0; --t) {
      size_t o = (size_t)t * (size_t)N + (size_t)i;
      float nd = term[o] ? 0.0f : 1.0f;
      if (mode == OIS_GAE_PPO) {
        float gi = g * nd;            /* discount_factor * not_dones[i] */
        float inner = nv + lam * a;   /* next_values + lambda_coefficient * advantage */
        a = (rew[o] - val[o]) + gi * inner;
      } else {
        float delta = (rew[o] + (g * nd) * nv) - val[o];
        a = delta + (gl * nd) * a;
      }
      adv[o] = a;
      ret[o] = a + val[o];            /* returns = advantages + values */
      nv = val[o];
      s1 += (double)a;
      s2 += (double)a * (double)a;
    }
  }
  if (normalize) {
    /* (advantages - advantages.mean()) / (advantages.std() + 1e-8), unbiased std */
    double n = (double)T * (double)N;
    double mean = s1 / n;
    double var = (s2 - n * mean * mean) / (n - 1.0);
    if (var < 0.0) var = 0.0;
    float mean_f = (float)mean;
    float den = (float)sqrt(var) + 1e-8f;
    for (size_t o = 0; o < (size_t)T * (size_t)N; ++o) adv[o] = (adv[o] - mean_f) / den;
  }
  return OIS_OK;
}
