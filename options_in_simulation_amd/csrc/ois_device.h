// ois_device.h — device-side building blocks of the fused skid-steer step kernel (gfx950).
//
// Execution shape ("quad layout"): 4 consecutive lanes = one arena, lane (tid & 3) = one
// wheel / one ground contact, in the env's wheel order fl, rl, fr, rr (KAMMA.py:157, 623).
// A wave64 therefore carries 16 arenas. Chassis quantities are replicated in the 4 lanes
// of a quad; per-wheel quantities (torque, spin, contact impulses) live in their lane.
// Cross-wheel sums are DPP quad_perm adds (no LDS, no ds_bpermute), cross-arena sums are
// wave butterflies. Every array in HBM is [N][4] so that lane tid reads element tid:
// each wave-level load/store instruction moves 256 contiguous bytes.
//
// Floating point: float32, operation order fixed by docs/SPEC.md; every fused multiply-add
// is an explicit __builtin_fmaf and the file is compiled with -ffp-contract=off, so results
// are bit-identical to the CPU oracle's.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ois.h"

namespace ois {

#define OIS_DEV __device__ __forceinline__

// ---------------------------------------------------------------- kernel constants ----
struct DevConsts {
  // physics (docs/SPEC.md §3)
  float dt, inv_dt, gdt, half_dt, im4, J4x, J4y, J4z, dt_over_Is, dt_over_Iy, rw, rw_over_Is,
      rw2_over_Is, beta_inv_dt, vdep, maxw, maxw2, damping, wx, wy, cz, com_height;
  int iters, nsub;
  // task (KAMMA.py constants, see include/ois.h)
  float Kp, Kp_omega, vehicle_mass, vehicle_inertia, policy_dt, action_scale, torque_clip, w1, w2,
      w3, sparse_bonus, sparse_threshold, terminal_reward, max_sim_steps, gamma1_const,
      gamma2_const, bounds, clip_obs, episode_length_s, eval_ramp_steps;
  float cmd_x_lo, cmd_x_hi, cmd_y_lo, cmd_y_hi, cmd_yaw_lo, cmd_yaw_hi, spawn_xy_range;
  float init_pos[3], init_quat[4], init_lin_vel[3], init_ang_vel[3], init_dof_vel[4];
  int action_dim, mix_mode, gamma1_mode, gamma2_mode, guiding_mode, cmd_sampler, eval_sums,
      max_episode_length;
  uint32_t seed_lo, seed_hi;
  int64_t arena_offset;
};

// device pointers of one environment ([N][4] unless noted)
struct DevBuffers {
  float *pose, *quat, *linvel, *angvel, *wheel_vel, *commands, *sums_a, *sums_b, *sums_c, *params,
      *lam_n, *lam_l, *lam_t, *wheel_torqs;
  int32_t* counters;
  float* obs;        // [N][6]
  float* rew;        // [N]
  uint8_t* done;     // [N]
  uint8_t* timeout;  // [N]
  float* partials;   // [nwaves][32] per-wave reset-event records
  int32_t* stamps;   // [nwaves]
  float* stat_acc;   // [nwaves][8]
  float* episode_info;  // [32]
  float* env0_diag;     // [32]
};

// ---------------------------------------------------------------- cross-lane ----------
template <int CTRL>
OIS_DEV float dpp(float x) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xF, 0xF, true));
}
template <int CTRL>
OIS_DEV int dppi(int x) {
  return __builtin_amdgcn_update_dpp(0, x, CTRL, 0xF, 0xF, true);
}
// (x0 + x1) + (x2 + x3) in every lane of the quad, bitwise identical in the 4 lanes
OIS_DEV float quad_sum(float x) {
  float s = x + dpp<0xB1>(x);  // quad_perm [1,0,3,2]
  return s + dpp<0x4E>(s);     // quad_perm [2,3,0,1]
}
template <int K>
OIS_DEV float quad_bcast(float x) { return dpp<K * 0x55>(x); }
template <int K>
OIS_DEV int quad_bcasti(int x) { return dppi<K * 0x55>(x); }
// component `w` of (a,b,c,d)
OIS_DEV float sel4(int w, float a, float b, float c, float d) {
  float lo = (w & 1) ? b : a, hi = (w & 1) ? d : c;
  return (w & 2) ? hi : lo;
}

// ---------------------------------------------------------------- RNG -----------------
// Philox4x32-10 (Salmon et al., SC'11; Random123 constants)
OIS_DEV void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                           uint32_t k1, uint32_t o[4]) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    uint32_t h0 = __umulhi(0xD2511F53u, c0), l0 = 0xD2511F53u * c0;
    uint32_t h1 = __umulhi(0xCD9E8D57u, c2), l1 = 0xCD9E8D57u * c2;
    uint32_t n0 = h1 ^ c1 ^ k0, n2 = h0 ^ c3 ^ k1;
    c0 = n0; c1 = l1; c2 = n2; c3 = l0;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  o[0] = c0; o[1] = c1; o[2] = c2; o[3] = c3;
}
OIS_DEV float u01(uint32_t x) { return (float)(x >> 8) * 5.9604644775390625e-08f; }
enum { PURPOSE_MASK = 0, PURPOSE_RESET_A = 1, PURPOSE_RESET_B = 2, PURPOSE_ACTION = 3, PURPOSE_PARAMS = 4 };
OIS_DEV void draw4(const DevConsts& K, int64_t gid, uint32_t step, uint32_t purpose, float u[4]) {
  uint32_t o[4];
  philox4x32_10((uint32_t)((uint64_t)gid & 0xffffffffu), (uint32_t)((uint64_t)gid >> 32), step, purpose,
                K.seed_lo, K.seed_hi, o);
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
OIS_DEV void rotT(const float R[9], const float a[3], float o[3]) {
  o[0] = __builtin_fmaf(R[6], a[2], __builtin_fmaf(R[3], a[1], R[0] * a[0]));
  o[1] = __builtin_fmaf(R[7], a[2], __builtin_fmaf(R[4], a[1], R[1] * a[0]));
  o[2] = __builtin_fmaf(R[8], a[2], __builtin_fmaf(R[5], a[1], R[2] * a[0]));
}
OIS_DEV void rot(const float R[9], const float a[3], float o[3]) {
  o[0] = __builtin_fmaf(R[2], a[2], __builtin_fmaf(R[1], a[1], R[0] * a[0]));
  o[1] = __builtin_fmaf(R[5], a[2], __builtin_fmaf(R[4], a[1], R[3] * a[0]));
  o[2] = __builtin_fmaf(R[8], a[2], __builtin_fmaf(R[7], a[1], R[6] * a[0]));
}

OIS_DEV float clipf(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }

// ---------------------------------------------------------------- per-lane state ------
struct Lane {
  // chassis (replicated across the quad)
  float p[3], vfwd;      // COM position, v_forward_projected of the previous step
  float q[4];
  float v[3], prev_lin;  // world linear velocity, previous_linear_velocity[:,0]
  float w[3], prev_ang;  // world angular velocity, previous_angular_velocity[:,2]
  float cmd[4];
  int ep_buf, sim_steps, ep_count;
  float mu, n[3];
  // per wheel (this lane's wheel)
  float jv;              // joint velocity (dof_vel)
  float ln, l1, l2;      // cached contact impulses
  float wt;              // clipped torque of this wheel (wheel_torqs)
  // lane-distributed episode sums: lane k holds component k of each float4 group
  float sa, sb, sc;
};

// One physics substep (docs/SPEC.md §3) for this lane's wheel. spin = absolute wheel spin.
OIS_DEV void substep(const DevConsts& K, int wl, float cx, float cy, Lane& S, float& spin) {
  float R[9];
  rotmat(S.q, R);
  float vb[3], wb[3], nb[3];
  rotT(R, S.v, vb);
  rotT(R, S.w, wb);
  rotT(R, S.n, nb);
  vb[0] = __builtin_fmaf(-K.gdt, R[6], vb[0]);
  vb[1] = __builtin_fmaf(-K.gdt, R[7], vb[1]);
  vb[2] = __builtin_fmaf(-K.gdt, R[8], vb[2]);
  // wheel drive, viscous joint damping, reaction on the chassis
  float jv = spin - wb[1];
  float T = __builtin_fmaf(-K.damping, jv, S.wt);
  spin = __builtin_fmaf(K.dt_over_Is, T, spin);
  float Tsum = quad_sum(T);
  wb[1] = __builtin_fmaf(-K.dt_over_Iy, Tsum, wb[1]);
  // contact frame (shared by the 4 wheels: common axle = body y)
  float s2 = __builtin_fmaf(nb[2], nb[2], nb[0] * nb[0]);
  s2 = fmaxf(s2, 1e-12f);
  float inv_s = 1.0f / sqrtf(s2);
  float dx = -(nb[0] * inv_s), dz = -(nb[2] * inv_s);
  float t1x = -dz, t1z = dx;
  float t2x = nb[1] * t1z;
  float t2y = __builtin_fmaf(nb[2], t1x, -(nb[0] * t1z));
  float t2z = -(nb[1] * t1x);
  float np = __builtin_fmaf(S.n[2], S.p[2], __builtin_fmaf(S.n[1], S.p[1], S.n[0] * S.p[0]));
  const float cz = K.cz;
  float rx = __builtin_fmaf(K.rw, dx, cx), ry = cy, rz = __builtin_fmaf(K.rw, dz, cz);
  float gap = np + __builtin_fmaf(nb[2], rz, __builtin_fmaf(nb[1], ry, nb[0] * rx));
  float b = gap >= 0.0f ? -(gap * K.inv_dt) : fminf(-(gap * K.beta_inv_dt), K.vdep);
  float anx = __builtin_fmaf(ry, nb[2], -(rz * nb[1]));
  float any = __builtin_fmaf(rz, nb[0], -(rx * nb[2]));
  float anz = __builtin_fmaf(rx, nb[1], -(ry * nb[0]));
  float a1x = cy * t1z;
  float a1y = __builtin_fmaf(cz, t1x, -(cx * t1z));
  float a1z = -(cy * t1x);
  float a2x = __builtin_fmaf(ry, t2z, -(rz * t2y));
  float a2y = __builtin_fmaf(rz, t2x, -(rx * t2z));
  float a2z = __builtin_fmaf(rx, t2y, -(ry * t2x));
  float Janx = K.J4x * anx, Jany = K.J4y * any, Janz = K.J4z * anz;
  float Ja1x = K.J4x * a1x, Ja1y = K.J4y * a1y, Ja1z = K.J4z * a1z;
  float Ja2x = K.J4x * a2x, Ja2y = K.J4y * a2y, Ja2z = K.J4z * a2z;
  float kn = __builtin_fmaf(anz, Janz, __builtin_fmaf(any, Jany, __builtin_fmaf(anx, Janx, K.im4)));
  float k1 = __builtin_fmaf(a1z, Ja1z, __builtin_fmaf(a1y, Ja1y, __builtin_fmaf(a1x, Ja1x, K.im4))) + K.rw2_over_Is;
  float k2 = __builtin_fmaf(a2z, Ja2z, __builtin_fmaf(a2y, Ja2y, __builtin_fmaf(a2x, Ja2x, K.im4)));
  float mn = 1.0f / kn, m1 = 1.0f / k1, m2 = 1.0f / k2;
  float vnx = K.im4 * nb[0], vny = K.im4 * nb[1], vnz = K.im4 * nb[2];
  float v1x = K.im4 * t1x, v1z = K.im4 * t1z;
  float v2x = K.im4 * t2x, v2y = K.im4 * t2y, v2z = K.im4 * t2z;
  float ln = S.ln, l1 = S.l1, l2 = S.l2;
  // warm start: one mass-split averaging pass with the cached impulses
  {
    float vx = __builtin_fmaf(l2, v2x, __builtin_fmaf(l1, v1x, __builtin_fmaf(ln, vnx, vb[0])));
    float vy = __builtin_fmaf(l2, v2y, __builtin_fmaf(ln, vny, vb[1]));
    float vz = __builtin_fmaf(l2, v2z, __builtin_fmaf(l1, v1z, __builtin_fmaf(ln, vnz, vb[2])));
    float ox = __builtin_fmaf(l2, Ja2x, __builtin_fmaf(l1, Ja1x, __builtin_fmaf(ln, Janx, wb[0])));
    float oy = __builtin_fmaf(l2, Ja2y, __builtin_fmaf(l1, Ja1y, __builtin_fmaf(ln, Jany, wb[1])));
    float oz = __builtin_fmaf(l2, Ja2z, __builtin_fmaf(l1, Ja1z, __builtin_fmaf(ln, Janz, wb[2])));
    spin = __builtin_fmaf(-K.rw_over_Is, l1, spin);
    vb[0] = 0.25f * quad_sum(vx); vb[1] = 0.25f * quad_sum(vy); vb[2] = 0.25f * quad_sum(vz);
    wb[0] = 0.25f * quad_sum(ox); wb[1] = 0.25f * quad_sum(oy); wb[2] = 0.25f * quad_sum(oz);
  }
  for (int it = 0; it < K.iters; ++it) {
    float vx = vb[0], vy = vb[1], vz = vb[2], ox = wb[0], oy = wb[1], oz = wb[2];
    // normal row
    float u = __builtin_fmaf(anz, oz, __builtin_fmaf(any, oy, __builtin_fmaf(anx, ox,
              __builtin_fmaf(nb[2], vz, __builtin_fmaf(nb[1], vy, nb[0] * vx)))));
    float dl = (b - u) * mn;
    float nw = fmaxf(ln + dl, 0.0f);
    dl = nw - ln;
    ln = nw;
    vx = __builtin_fmaf(dl, vnx, vx); vy = __builtin_fmaf(dl, vny, vy); vz = __builtin_fmaf(dl, vnz, vz);
    ox = __builtin_fmaf(dl, Janx, ox); oy = __builtin_fmaf(dl, Jany, oy); oz = __builtin_fmaf(dl, Janz, oz);
    float lim = S.mu * ln;
    // longitudinal row (couples the wheel spin)
    u = __builtin_fmaf(-K.rw, spin, __builtin_fmaf(a1z, oz, __builtin_fmaf(a1y, oy, __builtin_fmaf(a1x, ox,
        __builtin_fmaf(t1z, vz, t1x * vx)))));
    dl = -(u * m1);
    nw = fminf(fmaxf(l1 + dl, -lim), lim);
    dl = nw - l1;
    l1 = nw;
    vx = __builtin_fmaf(dl, v1x, vx); vz = __builtin_fmaf(dl, v1z, vz);
    ox = __builtin_fmaf(dl, Ja1x, ox); oy = __builtin_fmaf(dl, Ja1y, oy); oz = __builtin_fmaf(dl, Ja1z, oz);
    spin = __builtin_fmaf(-K.rw_over_Is, dl, spin);
    // lateral row
    u = __builtin_fmaf(a2z, oz, __builtin_fmaf(a2y, oy, __builtin_fmaf(a2x, ox,
        __builtin_fmaf(t2z, vz, __builtin_fmaf(t2y, vy, t2x * vx)))));
    dl = -(u * m2);
    nw = fminf(fmaxf(l2 + dl, -lim), lim);
    dl = nw - l2;
    l2 = nw;
    vx = __builtin_fmaf(dl, v2x, vx); vy = __builtin_fmaf(dl, v2y, vy); vz = __builtin_fmaf(dl, v2z, vz);
    ox = __builtin_fmaf(dl, Ja2x, ox); oy = __builtin_fmaf(dl, Ja2y, oy); oz = __builtin_fmaf(dl, Ja2z, oz);
    // mass-splitting average over the 4 contacts
    vb[0] = 0.25f * quad_sum(vx); vb[1] = 0.25f * quad_sum(vy); vb[2] = 0.25f * quad_sum(vz);
    wb[0] = 0.25f * quad_sum(ox); wb[1] = 0.25f * quad_sum(oy); wb[2] = 0.25f * quad_sum(oz);
  }
  S.ln = ln; S.l1 = l1; S.l2 = l2;
  // back to world, caps (origin_v18.py:65-70), integrate
  rot(R, vb, S.v);
  rot(R, wb, S.w);
  float w2 = __builtin_fmaf(S.w[2], S.w[2], __builtin_fmaf(S.w[1], S.w[1], S.w[0] * S.w[0]));
  if (w2 > K.maxw2) {
    float sc = K.maxw / sqrtf(w2);
    S.w[0] *= sc; S.w[1] *= sc; S.w[2] *= sc;
  }
  spin = fminf(fmaxf(spin, -K.maxw), K.maxw);
  S.p[0] = __builtin_fmaf(K.dt, S.v[0], S.p[0]);
  S.p[1] = __builtin_fmaf(K.dt, S.v[1], S.p[1]);
  S.p[2] = __builtin_fmaf(K.dt, S.v[2], S.p[2]);
  float qw = S.q[0], qx = S.q[1], qy = S.q[2], qz = S.q[3], h = K.half_dt;
  float nqw = __builtin_fmaf(-h, __builtin_fmaf(S.w[2], qz, __builtin_fmaf(S.w[1], qy, S.w[0] * qx)), qw);
  float nqx = __builtin_fmaf(h, __builtin_fmaf(-S.w[2], qy, __builtin_fmaf(S.w[1], qz, S.w[0] * qw)), qx);
  float nqy = __builtin_fmaf(h, __builtin_fmaf(-S.w[0], qz, __builtin_fmaf(S.w[2], qx, S.w[1] * qw)), qy);
  float nqz = __builtin_fmaf(h, __builtin_fmaf(-S.w[1], qx, __builtin_fmaf(S.w[0], qy, S.w[2] * qw)), qz);
  float nn = __builtin_fmaf(nqz, nqz, __builtin_fmaf(nqy, nqy, __builtin_fmaf(nqx, nqx, nqw * nqw)));
  float inv = 1.0f / sqrtf(nn);
  S.q[0] = nqw * inv; S.q[1] = nqx * inv; S.q[2] = nqy * inv; S.q[3] = nqz * inv;
}

// torch_rand_float(lower, upper) = (upper - lower) * rand + lower  (KAMMA.py:424-456)
OIS_DEV float rand_float(float lo, float hi, float u) { return (hi - lo) * u + lo; }

// a11: KAMMA.py:701-710
OIS_DEV float forward_projection(const float q[4], const float v[3]) {
  float f[3];
  quat_apply(q, 1.0f, 0.0f, 0.0f, f);
  float nrm = sqrtf((f[0] * f[0] + f[1] * f[1]) + f[2] * f[2]);
  float fx = f[0] / nrm, fy = f[1] / nrm, fz = f[2] / nrm;
  return (v[0] * fx + v[1] * fy) + v[2] * fz;
}

// What a resetting arena contributes to the per-wave reset-event record; lane k of the quad
// holds component k of every 4-vector.
struct ResetContrib {
  float v0, v1, v2, v3, v4, v5, v6;
};
OIS_DEV ResetContrib contrib_identity(int wl) {
  ResetContrib c;
  c.v0 = c.v1 = c.v2 = c.v3 = c.v4 = c.v5 = 0.0f;
  c.v6 = -__builtin_inff();
  return c;
}

// reset_idx for one arena (KAMMA.py:417-498); executed by the 4 lanes of the quad together.
OIS_DEV void reset_arena(const DevConsts& K, int wl, int64_t gid, float gamma1, float gamma2, Lane& S,
                         ResetContrib& rc) {
  float Sf = quad_bcast<3>(S.sb);  // "Final reward" sum
  float hg = 100.0f * (1.0f - gamma2);  // :470
  float gg = 100.0f * gamma2;           // :475
  float L = K.episode_length_s;
  // V0 = [count, ret^2, len, 0]
  rc.v0 = sel4(wl, 1.0f, Sf * Sf, (float)S.ep_buf, 0.0f);
  rc.v1 = S.sa;  // r1, r2, r3, dense
  rc.v2 = S.sb;  // sparse, guiding, observed, final
  rc.v3 = L * ((hg * S.sa) / Sf);  // :470-473
  float fb = (wl == 1) ? gg : hg;  // guiding uses gamma2 itself (:475)
  float rb = L * ((fb * S.sb) / Sf);  // :474-476
  rc.v4 = (wl == 3) ? 0.0f : rb;
  rc.v5 = (wl == 3) ? 0.0f : S.sc;  // smoothness, tracking error, desired velocity
  rc.v6 = sel4(wl, gamma1, gamma2, -Sf, Sf);
  S.sa = 0.0f; S.sb = 0.0f; S.sc = 0.0f;  // :485 (sc[3] = old-torque-valid flag, cleared too)

  float u[8];
  draw4(K, gid, (uint32_t)S.sim_steps, PURPOSE_RESET_A, &u[0]);
  draw4(K, gid, (uint32_t)S.sim_steps, PURPOSE_RESET_B, &u[4]);
  float r = K.spawn_xy_range;
  float px = K.init_pos[0] + rand_float(-r, r, u[0]);  // :422-424
  float py = K.init_pos[1] + rand_float(-r, r, u[1]);
  float pz = K.init_pos[2];
  float ang = rand_float(0.0f, 6.28318530717958647692f, u[2]);  // :426
  float s, c;
  sincos_det(ang / 2.0f, s, c);
  float qz[4] = {c, 0.0f, 0.0f, s};
  quat_mul(qz, K.init_quat, S.q);  // :434
  float cx = rand_float(K.cmd_x_lo, K.cmd_x_hi, u[3]);      // :448
  float cy = rand_float(K.cmd_y_lo, K.cmd_y_hi, u[4]);      // :451
  float ch = rand_float(K.cmd_yaw_lo, K.cmd_yaw_hi, u[5]);  // :454
  float keep = sqrtf(cx * cx + cy * cy) > 0.25f ? 1.0f : 0.0f;  // :458
  S.cmd[1] = cy * keep;
  S.cmd[3] = ch * keep;
  float x_vel = rand_float(K.cmd_x_lo, K.cmd_x_hi, u[6]);  // :571 (u[7]: discarded omega draw :572)
  if (K.cmd_sampler == OIS_CMD_BOX) {  // KAMMA_curr1.py:577-598
    float progress = fminf(1.0f, (float)S.sim_steps / K.max_sim_steps);
    float factor = 0.1f + (1.0f - 0.1f) * progress;
    float xc = (K.cmd_x_lo + K.cmd_x_hi) / 2.0f;
    float xmin = xc - (xc - K.cmd_x_lo) * factor;
    float xmax = xc + (K.cmd_x_hi - xc) * factor;
    x_vel = rand_float(xmin, xmax, u[6]);
  } else if (K.cmd_sampler == OIS_CMD_EVAL) {  // KAMMA_eval.py:545-550
    float progress = fminf(fmaxf((float)S.sim_steps / K.eval_ramp_steps, 0.0f), 1.0f);
    x_vel = 1.0f + progress * (2.0f - 1.0f);
  }
  S.cmd[0] = x_vel > 0.0f ? x_vel : 0.0f;  // :575, :495
  S.cmd[2] = 0.0f;                         // :574, :496
  float R[9];
  rotmat(S.q, R);
  S.p[0] = __builtin_fmaf(R[2], K.com_height, px);
  S.p[1] = __builtin_fmaf(R[5], K.com_height, py);
  S.p[2] = __builtin_fmaf(R[8], K.com_height, pz);
  S.v[0] = K.init_lin_vel[0]; S.v[1] = K.init_lin_vel[1]; S.v[2] = K.init_lin_vel[2];  // :435
  S.w[0] = K.init_ang_vel[0]; S.w[1] = K.init_ang_vel[1]; S.w[2] = K.init_ang_vel[2];
  S.jv = sel4(wl, K.init_dof_vel[0], K.init_dof_vel[1], K.init_dof_vel[2], K.init_dof_vel[3]);  // :438
  S.ln = 0.0f; S.l1 = 0.0f; S.l2 = 0.0f;
  S.ep_buf = 0;    // :462
  S.ep_count += 1; // :463
}

// Butterfly over the 16 arenas of the wave (lane bits 2..5); lane k keeps component k.
template <int OP>
OIS_DEV float wave_reduce16(float x) {
#pragma unroll
  for (int m = 4; m < 64; m <<= 1) {
    float y = __shfl_xor(x, m, 64);
    x = OP == 0 ? x + y : fmaxf(x, y);
  }
  return x;
}

// Per-wave record of a reset event + finished-episode accumulators. Must be called by all
// 64 lanes of the wave. `any` = wave has at least one resetting arena.
OIS_DEV void wave_record(const DevBuffers& B, int wave, int lane, const ResetContrib& rc, int stamp,
                         bool count_stats) {
  float r0 = wave_reduce16<0>(rc.v0), r1 = wave_reduce16<0>(rc.v1), r2 = wave_reduce16<0>(rc.v2);
  float r3 = wave_reduce16<0>(rc.v3), r4 = wave_reduce16<0>(rc.v4), r5 = wave_reduce16<0>(rc.v5);
  float r6 = wave_reduce16<1>(rc.v6);
  if (lane < 4) {
    float* P = B.partials + (size_t)wave * 32;
    P[0 + lane] = r0; P[4 + lane] = r1; P[8 + lane] = r2; P[12 + lane] = r3;
    P[16 + lane] = r4; P[20 + lane] = r5; P[24 + lane] = r6;
    if (lane == 0) B.stamps[wave] = stamp;
    if (count_stats) {
      float* A = B.stat_acc + (size_t)wave * 8;
      // A[0..3] = count, ret^2, len, ret ; A[4..7] = -, -, max(-ret), max(ret)
      A[lane] += (lane == 3) ? r2 : r0;
      if (lane >= 2) A[4 + lane] = fmaxf(A[4 + lane], r6);
    }
  }
}

}  // namespace ois
