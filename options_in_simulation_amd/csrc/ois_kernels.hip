// ois_kernels.hip — fused policy-step kernel for N skid-steer arenas on gfx950 + the C ABI
// of include/ois.h. Reference path: TorqueDistributionTask.pre_physics_step /
// post_physics_step (src/my_envs/KAMMA.py:596-727) with PhysX replaced by the build-defined
// rigid-body model of docs/SPEC.md §3.
//
// One launch = one (or k) policy step(s) for all arenas: controller -> action mixing ->
// torque clip -> 20 contact-solver substeps -> termination -> reward -> auto-reset ->
// observation -> forward-speed projection, state in registers throughout.
// step_body<LPA, PK, TER, NIT> behind two kernel shells (step_kernel on the plane, step_kernel_terrain
// on a height field): LPA lanes per arena (1 = a lane owns an arena, 2 = a lane pair
// shares one), PK packed fp32 over wheel pairs (ois_physics_pk.h), TER wheel contact on a height
// field, NIT solver iterations unrolled at compile time (0 = run-time count). Also here:
// reset / params / episode-info / stats / height-scan kernels; ois_gae.hip holds the GAE scan.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <new>

#include "ois_physics_pk.h"

using namespace ois;

namespace {

#ifndef OIS_BLOCK
#define OIS_BLOCK 64  // threads per workgroup of the step / reset kernels (one wave)
#endif
constexpr int kBlock = OIS_BLOCK;

// ------------------------------------------------------------------ load / store ------
OIS_DEV void unpack(const float4 t, float* d) { d[0] = t.x; d[1] = t.y; d[2] = t.z; d[3] = t.w; }
// State rows are written once per launch and next read by the NEXT launch, after the caller's policy has swept
// the caches: non-temporal stores let them drain while the wave is still finishing instead of sitting dirty in
// L2 until the end-of-kernel write-back (single ois_step launch 34.10 -> 33.65 us back to back, 36.11 -> 35.45
// behind a cache sweep, tools/inloop_probe.py). Same values, same bytes.
typedef float v4f_t __attribute__((ext_vector_type(4)));
typedef int v4i_t __attribute__((ext_vector_type(4)));
typedef unsigned v2u_t __attribute__((ext_vector_type(2)));
OIS_DEV void store_row(float4* p, float a, float b, float c, float d) {
  const v4f_t v = {a, b, c, d};
  __builtin_nontemporal_store(v, reinterpret_cast<v4f_t*>(p));
}
OIS_DEV void store_row(int4* p, int a, int b, int c, int d) {
  const v4i_t v = {a, b, c, d};
  __builtin_nontemporal_store(v, reinterpret_cast<v4i_t*>(p));
}
template <int LPA>
OIS_DEV void load_wheels(const float4* base, int arena, int h, float* d) {  // this lane's wheels of a [N][4] row
  if (LPA == 1) { unpack(base[arena], d); }
  else { const float2 t = reinterpret_cast<const float2*>(base)[arena * 2 + h]; d[0] = t.x; d[1] = t.y; }
}
template <int LPA>
OIS_DEV void store_wheels(float4* base, int arena, int h, const float* s) {
  if (LPA == 1) store_row(&base[arena], s[0], s[1], s[2], s[3]);
  else reinterpret_cast<float2*>(base)[arena * 2 + h] = make_float2(s[0], s[1]);
}

// the warm-start impulses: 4 halves per arena and row (this lane's 4 / LPA of them)
template <int LPA>
OIS_DEV void load_lambda(const uint2* base, int arena, int h, float* d) {
  if (LPA == 1) { const uint2 t = base[arena]; unpack_half2(t.x, d[0], d[1]); unpack_half2(t.y, d[2], d[3]); }
  else { unpack_half2(reinterpret_cast<const uint32_t*>(base)[arena * 2 + h], d[0], d[1]); }
}
template <int LPA>
OIS_DEV void store_lambda(uint2* base, int arena, int h, const float* s) {
  if (LPA == 1) { const v2u_t v = {pack_half2(s[0], s[1]), pack_half2(s[2], s[3])}; __builtin_nontemporal_store(v, reinterpret_cast<v2u_t*>(&base[arena])); }
  else reinterpret_cast<uint32_t*>(base)[arena * 2 + h] = pack_half2(s[0], s[1]);
}

template <int LPA>
OIS_DEV void load_arena(bool eval_sums, const DevBuffers& B, int i, int h, ArenaT<LPA>& S) {
  // issued in the order of first use (vector loads return in order, so a later row never holds up an
  // earlier one): controller inputs, then what the first substep reads, the episode sums last
  // (only read after the 20 substeps)
  const float4 cm = B.commands[i], po = B.pose[i], av = B.angvel[i];
  const int4 ct = B.counters[i];
  const float4 qu = B.quat[i], lv = B.linvel[i], pr = B.params[i];
  unpack(cm, S.cmd);
  S.p[0] = po.x; S.p[1] = po.y; S.p[2] = po.z; S.vfwd = po.w;
  S.w[0] = av.x; S.w[1] = av.y; S.w[2] = av.z; S.prev_ang = av.w;
  S.ep_buf = ct.x; S.sim_steps = ct.y; S.ep_count = ct.z;
  unpack(qu, S.q);
  S.v[0] = lv.x; S.v[1] = lv.y; S.v[2] = lv.z; S.prev_lin = lv.w;
  S.mu = pr.x; S.n[0] = pr.y; S.n[1] = pr.z; S.n[2] = pr.w;
  load_wheels<LPA>(B.wheel_vel, i, h, S.jv);
  load_lambda<LPA>(B.lam_n, i, h, S.ln);
  load_lambda<LPA>(B.lam_l, i, h, S.l1);
  load_lambda<LPA>(B.lam_t, i, h, S.l2);
  if (eval_sums) {
    unpack(B.sums_c[i], S.sc);
    load_wheels<LPA>(B.wheel_torqs, i, h, S.wt);
  } else {
    for (int k = 0; k < 4; ++k) S.sc[k] = 0.0f;
    for (int k = 0; k < ArenaT<LPA>::WPL; ++k) S.wt[k] = 0.0f;
  }
  const float4 sa = B.sums_a[i], sb = B.sums_b[i];
  unpack(sa, S.sa); unpack(sb, S.sb);
}

template <int LPA>
OIS_DEV void store_arena(bool eval_sums, const RowPtrs& R, const WheelPtrs& Wp, int i, int h, const ArenaT<LPA>& S, bool cmd_dirty) {
  if (h == 0) {  // chassis rows are replicated in the lane pair: one lane writes them
    store_row(&R.pose[i], S.p[0], S.p[1], S.p[2], S.vfwd);
    store_row(&R.quat[i], S.q[0], S.q[1], S.q[2], S.q[3]);
    store_row(&R.linvel[i], S.v[0], S.v[1], S.v[2], S.prev_lin);
    store_row(&R.angvel[i], S.w[0], S.w[1], S.w[2], S.prev_ang);
    if (cmd_dirty) R.commands[i] = make_float4(S.cmd[0], S.cmd[1], S.cmd[2], S.cmd[3]);  // only reset_idx writes commands
    store_row(&R.counters[i], S.ep_buf, S.sim_steps, S.ep_count, 0);
    store_row(&R.sums_a[i], S.sa[0], S.sa[1], S.sa[2], S.sa[3]);
    store_row(&R.sums_b[i], S.sb[0], S.sb[1], S.sb[2], S.sb[3]);
    if (eval_sums) Wp.sums_c[i] = make_float4(S.sc[0], S.sc[1], S.sc[2], S.sc[3]);
  }
  store_wheels<LPA>(Wp.wheel_vel, i, h, S.jv);
  store_lambda<LPA>(Wp.lam_n, i, h, S.ln);  // already binary16 values (rounded at the end of the policy step)
  store_lambda<LPA>(Wp.lam_l, i, h, S.l1);
  store_lambda<LPA>(Wp.lam_t, i, h, S.l2);
  store_wheels<LPA>(Wp.wheel_torqs, i, h, S.wt);
}

// ------------------------------------------------------------------ the step kernel ---
template <int LPA, bool PK, bool TER, int NIT, bool FLAT = false, bool FOLD = false, int TSTRIDE = 1>
OIS_DEV void step_body(const PhysK& K, const DevBuffers& B, const float* __restrict__ actions,
                       float* __restrict__ reward_sum, int N, int ksteps, int policy, int rp_head) {
  // K: the physics constants only, by value. Everything else of the kernel's arguments that a LATER phase needs (task
  // constants, reset constants, the pointers of the stores) is re-read from the kernarg segment where that phase starts
  // (kernarg_late, ois_device.h) and never touched through the by-value argument: what is not live across the substep
  // loop is not spilled around it.
  // [phase: prologue: argument blocks, state rows in (per launch)]
  constexpr int W = 4 / LPA;
  const int tid = blockIdx.x * kBlock + threadIdx.x;
  const int lane = threadIdx.x & 63;
  const int arena = tid / LPA;
  const int h = LPA == 1 ? 0 : (tid & 1);  // which side of the arena this lane owns
  const int wheel0 = h * W;
  const bool valid = arena < N;
  const int la = valid ? arena : N - 1;  // tail lanes shadow the last arena and store nothing
  kernarg_t kp = kernarg_base();
  const TaskK T0 = late_task(kp);  // the prologue's own copy
  const int64_t gid = T0.arena_offset + la;
  const bool eval_sums = T0.eval_sums != 0;

  ArenaT<LPA> S;
  load_arena<LPA>(eval_sums, B, la, h, S);
  // bit 8 of `policy`: this launch continues a rollout that is split into launches (OIS_CMD_LEVEL)
  const bool rsum_continue = (policy & 0x100) != 0;
  // bit 9: the per-level command re-draw that follows the PREVIOUS step of this rollout (level_command_kernel's work,
  // KAMMA.py:719-725) is done here, in the prologue of the next step's launch, instead of by a launch of its own
  // (FOLD shells only: carried by every shell, the code cost the default path 0.2 us per launch, profiles/r3_ab_fold.txt)
  const bool fold_level = FOLD && (policy & 0x200) != 0;
  policy &= 0xff;
  float rsum = (rsum_continue && reward_sum != nullptr) ? reward_sum[la] : 0.0f;
  bool cmd_dirty = false;
  // KAMMA_eval.py:470: a reset_idx call zeroes old_torques of EVERY arena. Did any arena reset in the step before
  // this launch? (the stamp slot of this parity was written by that step or by an older one; launches of the
  // eval variants are single steps, launch_steps())
  const bool old_zeroed = eval_sums && B.reset_stamp[S.sim_steps & 1] == S.sim_steps;
  // terrain tile of this arena (origin xyz; level / type for the curriculum)
  float tor[3] = {0.0f, 0.0f, 0.0f};
  int tlevel = 0, ttype = 0;
  if constexpr (TER) {
    const float4 eo = B.env_origins[la];
    const int4 ti = B.terrain_idx[la];
    tor[0] = eo.x; tor[1] = eo.y; tor[2] = eo.z;
    tlevel = ti.x; ttype = ti.y;
  }
  // [census: not in the bench configuration: continuation launch of a level-curriculum rollout]
  if constexpr (FOLD) if (fold_level) {
    const ResetK Rk = late_reset(kp);
    const int sim = S.sim_steps;  // == episode_step_counter of the step just done
    float fr = B.cmd_fraction[la];
    if (B.reset_stamp[sim & 1] == sim) {  // some arena reset in that step: `fraction` of every arena is re-snapshot (:498)
      fr = S.sa[0] / Rk.cmd_thr_high;
      if (valid && h == 0) B.cmd_fraction[la] = fr;
    }
    float u[4];
    draw4(T0, gid, (uint32_t)sim, PURPOSE_CMD, u);
    level_command(TER ? tlevel : 0, fr, sim, normal_det(u[0], u[1]), Rk.cmd_dt, Rk.cmd_episode_s, S.cmd[0], S.cmd[2]);
    cmd_dirty = true;
  }
  // [census: end]
  // FIFO transition memory: "states" of the first appended row = the observation the caller saw
  float o_prev[6] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
  if (B.rp_size > 0) {
    const float* o = B.obs + (size_t)la * 6;
    const float2 t0 = *reinterpret_cast<const float2*>(o), t1 = *reinterpret_cast<const float2*>(o + 2),
                 t2 = *reinterpret_cast<const float2*>(o + 4);
    o_prev[0] = t0.x; o_prev[1] = t0.y; o_prev[2] = t1.x; o_prev[3] = t1.y; o_prev[4] = t2.x; o_prev[5] = t2.y;
  }

  for (int s = 0; s < ksteps; ++s) {
    const bool last = s == ksteps - 1;
    const TaskK T = late_task(kp);  // pre_physics_step's copy: dead before the first substep
    const int A = T.action_dim;
    // [phase: actions in (per step)]
    // ---- actions -> this lane's wheels (KA_DDPG_2D.py:623-624, KA_DDPG_1D.py:600-603) ----
    float a4[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    if (policy == OIS_POLICY_EXTERNAL) {
      const float* ap = actions + ((size_t)s * N + la) * A;
      if (A == 4) { unpack(*reinterpret_cast<const float4*>(ap), a4); }
      else if (A == 2) { const float2 t = *reinterpret_cast<const float2*>(ap); a4[0] = a4[1] = t.x; a4[2] = a4[3] = t.y; }
      else { a4[0] = a4[1] = a4[2] = a4[3] = ap[0]; }
    // [census: not in the bench configuration: in-kernel random policy]
    } else if (policy == OIS_POLICY_RANDOM) {
      float u[4];
      draw4(T, gid, (uint32_t)S.sim_steps, PURPOSE_ACTION, u);
      if (A == 4) { a4[0] = u[0]; a4[1] = u[1]; a4[2] = u[2]; a4[3] = u[3]; }
      else if (A == 2) { a4[0] = a4[1] = u[0]; a4[2] = a4[3] = u[1]; }
      else { a4[0] = a4[1] = a4[2] = a4[3] = u[0]; }
    }
    // [census: end]
    float aw[W];
#pragma unroll
    for (int k = 0; k < W; ++k) {
      // value selects, never an index computed from h: a run-time index would make the compiler
      // move a4 to LDS, and that code reads the workgroup size through the dispatch pointer —
      // host memory, ~13 us per launch (tools/lds_launch_probe.hip)
      const float lo = a4[k & 3], hi = a4[(2 + k) & 3];
      aw[k] = (LPA == 1) ? lo : (h ? hi : lo);
    }
    // [phase: pre_physics_step: controller, gamma, switch draw, mix, clip (per step)]
    // ---- pre_physics_step (KAMMA.py:596-656) ----
    const float current_v = S.vfwd;      // :607
    const float current_omega = S.w[2];  // :608
    const float vd = S.cmd[0] - current_v;      // :609
    const float od = S.cmd[2] - current_omega;  // :610
    const float lin = T.Kp * (T.vehicle_mass * (vd / T.policy_dt));            // :617
    const float ang = T.Kp_omega * (T.vehicle_inertia * (od / T.policy_dt));
    const float ac_left = lin - ang, ac_right = lin + ang;                     // :617, :619
    float gs = 1.0f - ((float)S.sim_steps / T.max_sim_steps);                  // :627-628
    gs = gs < 0.0f ? 0.0f : gs;
    const float gamma1 = T.gamma1_mode == OIS_GAMMA_CONST ? T.gamma1_const : gs;
    const float gamma2 = T.gamma2_mode == OIS_GAMMA_CONST ? T.gamma2_const : gs;
    float rnd = 0.0f;
    if (T.mix_mode == OIS_MIX_SWITCH) {
      float u[4];
      draw4(T, gid, (uint32_t)S.sim_steps, PURPOSE_MASK, u);  // torch.rand(num_envs, 1) :633
      rnd = u[0];
    }
    float torq[W], wt[W], pol[W], d2[W], w2s[W];
#pragma unroll
    for (int k = 0; k < W; ++k) {
      const bool left = (LPA == 1) ? (k < 2) : (h == 0);
      const float crit = left ? ac_left : ac_right;  // :624 order fl, rl, fr, rr
      pol[k] = aw[k] * T.action_scale;
      float t;
      if (T.mix_mode == OIS_MIX_SWITCH) {
        const float mask = rnd < gamma1 ? 1.0f : 0.0f;  // :634
        t = mask * crit + (1.0f - mask) * pol[k];        // :635
      } else {
        t = ((1.0f - gamma1) * aw[k]) * T.action_scale + gamma1 * crit;  // KA_DDPG_4D.py:631
      }
      torq[k] = t;
      wt[k] = clipf(t, -T.torque_clip, T.torque_clip);  // :649
      const float d = (T.guiding_mode == OIS_GUIDE_EXECUTED ? wt[k] : pol[k]) - crit;
      d2[k] = d * d;
      w2s[k] = wt[k] * wt[k];
    }
    const float guiding = -sqrtf(sum_wheels<LPA>(d2));  // :656 / KA_DDPG_4D.py:637
    // [census: not in the bench configuration: eval variants]
    if (T.eval_sums) {  // KAMMA_eval.py:677-685
      float e2[W];
#pragma unroll
      for (int k = 0; k < W; ++k) {
        const float old = (S.sc[3] != 0.0f && !(s == 0 && old_zeroed)) ? S.wt[k] : 0.0f;
        const float df = old - wt[k];
        e2[k] = df * df;
      }
      S.sc[0] += sum_wheels<LPA>(e2) / 10.0f;
      S.sc[2] += S.cmd[0] / 10.0f;
      S.sc[3] = 1.0f;
    }
    // [census: end]
#pragma unroll
    for (int k = 0; k < W; ++k) S.wt[k] = wt[k];

    // [phase: substeps (the first one exact, the loop; + body <-> world glue)]
    // ---- physics: decimation (:646-653) + controlFrequencyInv (:43) substeps ----
    float R[9];
    bool blown = false;
    const int kp_parked = kernarg_park(kp);
    {
      PhysT<LPA> X;
      float wb[3];
      rotmat(S.q, R);
      rotT(R, S.w, wb);
#pragma unroll
      for (int k = 0; k < 3; ++k) { X.p[k] = S.p[k]; X.v[k] = S.v[k]; X.wb[k] = wb[k]; X.n[k] = S.n[k]; }
#pragma unroll
      for (int k = 0; k < 4; ++k) X.q[k] = S.q[k];
#pragma unroll
      for (int k = 0; k < W; ++k) {
        X.tau[k] = wt[k]; X.spin[k] = S.jv[k] + wb[1];
        X.ln[k] = S.ln[k]; X.l1[k] = S.l1[k]; X.l2[k] = S.l2[k];
      }
      X.mu = S.mu;
#pragma unroll
      for (int k = 0; k < 3; ++k) X.to[k] = tor[k];
      if constexpr (TER && TSTRIDE == 2) {
        // ois_terrain.footprint_stride = 2 (SPEC 3.8): the later substeps in pairs — an odd one fits the plane to the stored
        // sample and samples nothing, the even one after it samples and keeps that plane
        int i = 1;
        if constexpr (PK) {
          substep_pk<true, true, NIT>(K, B.ter, X);
          for (; i + 1 < K.nsub; i += 2) {
            substep_pk<false, true, NIT, false, kOddSubstep>(K, B.ter, X);
            substep_pk<false, true, NIT, false, kEvenSubstep>(K, B.ter, X);
          }
          if (i < K.nsub) substep_pk<false, true, NIT, false, kOddSubstep>(K, B.ter, X);
        } else {
          substep<true, 1, true, NIT>(K, B.ter, wheel0, X);
          for (; i + 1 < K.nsub; i += 2) {
            substep<false, 1, true, NIT, kOddSubstep>(K, B.ter, wheel0, X);
            substep<false, 1, true, NIT, kEvenSubstep>(K, B.ter, wheel0, X);
          }
          if (i < K.nsub) substep<false, 1, true, NIT, kOddSubstep>(K, B.ter, wheel0, X);
        }
      } else if constexpr (PK && LPA == 1) {  // packed fp32 over wheel pairs (ois_physics_pk.h)
        substep_pk<true, TER, NIT, FLAT>(K, B.ter, X);
        for (int i = 1; i < K.nsub; ++i) substep_pk<false, TER, NIT, FLAT>(K, B.ter, X);
      } else if constexpr (PK && LPA == 2) {  // a lane pair per arena, each lane one packed side pair
        substep_pk2<true, NIT, FLAT>(K, wheel0, X);
        for (int i = 1; i < K.nsub; ++i) substep_pk2<false, NIT, FLAT>(K, wheel0, X);
      } else {
        substep<true, LPA, TER, NIT>(K, B.ter, wheel0, X);
        for (int i = 1; i < K.nsub; ++i) substep<false, LPA, TER, NIT>(K, B.ter, wheel0, X);
      }
      // [phase: after the substeps: binary16 rounding, finite check, rotation back (per step)]
      // safety net: a non-finite outcome (diverged reciprocal refresh of a violently tumbling
      // robot) is discarded: pre-step pose at rest, reset as fallen (oracle step_arena, SPEC §3.9)
      // the impulses cross a policy-step boundary in binary16, inside a fused rollout too (SPEC 3.5);
      // rounded before the check, so an impulse beyond the half range (inf) counts as non-finite
#pragma unroll
      for (int k = 0; k < W; ++k) { X.ln[k] = round_half(X.ln[k]); X.l1[k] = round_half(X.l1[k]); X.l2[k] = round_half(X.l2[k]); }
      float chk = 0.0f;
#pragma unroll
      for (int k = 0; k < 3; ++k) chk += fabsf(X.p[k]) + fabsf(X.v[k]) + fabsf(X.wb[k]);
#pragma unroll
      for (int k = 0; k < 4; ++k) chk += fabsf(X.q[k]);
      float cw[W];
#pragma unroll
      for (int k = 0; k < W; ++k) cw[k] = fabsf(X.spin[k]) + fabsf(X.ln[k]) + fabsf(X.l1[k]) + fabsf(X.l2[k]);
      chk += sum_wheels<LPA>(cw);
      // ... or a linear speed beyond 50 m/s: the finite garbage of a Newton refresh in a singular pose (oracle step_arena, SPEC 3.9)
      const float v2 = __builtin_fmaf(X.v[2], X.v[2], __builtin_fmaf(X.v[1], X.v[1], X.v[0] * X.v[0]));
      blown = !(chk < 3.0e38f) || !(v2 < 2500.0f);
      // A blown arena is reset below (reset |= blown), and reset_arena overwrites every state value the oracle restores or
      // zeroes for it here (pose, velocities, wheel speeds, impulses: oracle step_arena). What its metrics still see of
      // this step's outcome is the yaw rate alone (ang_acc, prev_ang): zeroed after the rotation. The rest may hold NaN
      // until reset_arena runs; no other lane reads it. (The per-value restore cost 54 register moves on every step.)
      rotmat(X.q, R);
      rot(R, X.wb, S.w);  // body -> world with the final orientation
      S.w[2] = blown ? 0.0f : S.w[2];
#pragma unroll
      for (int k = 0; k < 3; ++k) { S.p[k] = X.p[k]; S.v[k] = X.v[k]; }
#pragma unroll
      for (int k = 0; k < 4; ++k) S.q[k] = X.q[k];
#pragma unroll
      for (int k = 0; k < W; ++k) {
        S.jv[k] = X.spin[k] - X.wb[1];
        S.ln[k] = X.ln[k]; S.l1[k] = X.l1[k]; S.l2[k] = X.l2[k];
      }
    }

    // [phase: post_physics_step: is_done, calculate_metrics (per step)]
    // ---- post_physics_step (:674-727) ----
    kp = kernarg_unpark(kp_parked);
    const TaskK T2 = late_task(kp);  // its own copy: nothing of the task constants is held across the substeps
    S.ep_buf += 1;     // :675
    S.sim_steps += 1;  // :676
    // base-link position for the bounds check (:758-759)
    const float plx = __builtin_fmaf(-R[2], T2.com_height, S.p[0]);
    const float ply = __builtin_fmaf(-R[5], T2.com_height, S.p[1]);
    // is_done (:736-764)
    const bool timeout = S.ep_buf >= T2.max_episode_length - 1;  // :740-744
    float pg[3];
    quat_apply(S.q, 0.0f, 0.0f, -1.0f, pg);                      // :749
    const bool fallen = pg[2] > 0.0f;                            // :753
    const bool oob = (plx < -T2.bounds) | (plx > T2.bounds) | (ply < -T2.bounds) | (ply > T2.bounds);
    const bool reset = fallen | timeout | oob | blown;
    // calculate_metrics (:767-822)
    const float lin_acc = (current_v - S.prev_lin) / T2.policy_dt;  // :770
    const float ang_acc = (S.w[2] - S.prev_ang) / T2.policy_dt;     // :771
    S.prev_lin = current_v;                                        // :774
    S.prev_ang = S.w[2];                                           // :775
    const float r1 = vd * vd + 5.0f * (od * od);                   // :779
    const float r2 = lin_acc * lin_acc + ang_acc * ang_acc;        // :781
    const float r3 = sum_wheels<LPA>(w2s);                         // :783
    const float c1 = T2.w1 * r1, c2 = T2.w2 * r2, c3 = T2.w3 * r3;
    const float rdense = (c1 + c2) + c3;                           // :786
    const float sparse = (fabsf(vd) < T2.sparse_threshold && fabsf(od) < T2.sparse_threshold) ? T2.sparse_bonus : 0.0f;
    const float observed = rdense + sparse;                        // :795
    float rew = (1.0f - gamma2) * observed + gamma2 * guiding;     // :798
    rew += (T2.terminal_reward * (reset ? 1.0f : 0.0f)) * (timeout ? -2.0f : -1.0f);  // :801 (~ on an int64 tensor)
    S.sa[0] += c1; S.sa[1] += c2; S.sa[2] += c3; S.sa[3] += rdense;               // :803-806
    S.sb[0] += sparse; S.sb[1] += guiding; S.sb[2] += observed; S.sb[3] += rew;   // :807-810
    if (T2.eval_sums) S.sc[1] += fabsf(vd) / 10.0f;                 // KAMMA_eval.py:840
    // [phase: reset_idx (rare: a lane that resets)]
    // reset_idx (:417-498)
    ResetContrib rc;
    contrib_identity(rc);
    const bool do_reset = reset && valid;
    if constexpr (TER) {
      if (reset && B.ter.curriculum) {
        // update_SI_level (KAMMA.py:502-529) on the finished episode's r1 sum, then the new tile origin
        if (S.sa[0] < B.ter.thr_lo) tlevel = tlevel - 1 < 0 ? 0 : tlevel - 1;
        if (S.sa[0] > B.ter.thr_hi) tlevel = tlevel + 1 > B.ter.max_level ? B.ter.max_level : tlevel + 1;
        const float* o = B.ter.origins + ((size_t)tlevel * B.ter.ntypes + ttype) * 3;
        tor[0] = o[0]; tor[1] = o[1]; tor[2] = o[2];  // :529
      }
    }
    if (reset) {
      const ResetK Rk = late_reset(kp);
      reset_arena<LPA, !FLAT && !TER>(T2, Rk, gid, wheel0, gamma1, gamma2, S, rc);
    }
    cmd_dirty |= reset;
    if (!valid) contrib_identity(rc);
    // [phase: get_observations (per step)]
    // get_observations (:833); desired_* are views of commands -> post-reset values; clamp: rl_task.py:99
    const float co = T2.clip_obs;
    const float ob[6] = {clipf(S.cmd[0], -co, co), clipf(S.cmd[2], -co, co), clipf(vd, -co, co),
                         clipf(od, -co, co),       clipf(lin_acc, -co, co),  clipf(ang_acc, -co, co)};
    // KAMMA_g1d_g21.py:838-842: get_observations replaces a (nearly) zero rew_buf[0] by 1e-12 before it is
    // returned; the episode sums above already hold the unmodified value. Global arena 0: a sharded run equals the
    // unsharded one (the env-0 diagnostics below stay per environment, i.e. per shard)
    if (T2.env0_reward_guard && gid == 0 && fabs((double)rew) < 1e-12) rew = 1e-12f;
    rsum += rew;
    // [phase: transition memory row (only with a memory attached)]
    // FIFO transition memory row (train_kaddpg_4d_nofifo_random_RLIL.py:147-156; executed action:
    // base_KA_DDPG.py:212)
    if (B.rp_size > 0 && valid) {
      const AuxPtrs X = kernarg_late<AuxPtrs, kArgB + (int)offsetof(DevBuffers, rp_states)>(kp);
      const size_t row = (size_t)((rp_head + s) % B.rp_size) * N + arena;
#pragma unroll
      for (int k = 0; k < W; ++k) X.rp_actions[row * 4 + wheel0 + k] = wt[k] / T2.action_scale;
      if (h == 0) {
#pragma unroll
        for (int k = 0; k < 6; ++k) { X.rp_states[row * 6 + k] = o_prev[k]; X.rp_next[row * 6 + k] = ob[k]; }
        X.rp_rewards[row] = rew;
        X.rp_done[row] = reset ? 1 : 0;
      }
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) o_prev[k] = ob[k];
    // [phase: outputs and env-0 diagnostics (last step of a launch)]
    // outputs (only the last step of a rollout is observable)
    // (Sending the [N, 6] observation rows through an LDS tile so that they leave as whole 128-byte lines — one float4
    // per lane instead of three float2 at a 24-byte stride — was built and measured in round 3: 9.42 vs 8.96 us for a
    // one-substep launch back to back, 36.25 vs 35.88 behind a cache sweep (profiles/r3_ab_flat_transposed.txt): the
    // barrier and the LDS round trip cost more than the partial-line stores, which L2 merges anyway. Not kept.)
    if (last && valid) {
      const OutPtrs O = kernarg_late<OutPtrs, kArgB + (int)offsetof(DevBuffers, obs)>(kp);
      if (h == 0) {
        float* o = O.obs + (size_t)arena * 6;
        *reinterpret_cast<float2*>(o) = make_float2(ob[0], ob[1]);
        *reinterpret_cast<float2*>(o + 2) = make_float2(ob[2], ob[3]);
        *reinterpret_cast<float2*>(o + 4) = make_float2(ob[4], ob[5]);
        O.rew[arena] = rew;
        O.done[arena] = reset ? 1 : 0;
        O.timeout[arena] = timeout ? 1 : 0;
      }
      if (arena == 0) {  // task.observed_components ingredients (:840-867)
        float* d = O.env0_diag;
        for (int k = 0; k < W; ++k) {
          const int gk = wheel0 + k;
          const int ti = gk < 2 ? 9 + gk : 11 + (gk - 2);  // torque_apl fl rl | fr rr at 9,10,11,12
          d[ti] = torq[k];
          d[15 + gk] = pol[k];
        }
        if (h == 0) {
          d[0] = S.cmd[0]; d[1] = current_v; d[2] = S.cmd[2]; d[3] = current_omega; d[4] = vd; d[5] = od;
          d[6] = lin_acc; d[7] = ang_acc; d[8] = (float)S.ep_count;
          d[13] = ac_left; d[14] = ac_right;
          d[19] = gamma2; d[20] = c1; d[21] = c2; d[22] = c3; d[23] = rdense; d[24] = sparse; d[25] = observed;
          d[26] = guiding; d[28] = gamma1;
          // the percentages of observed_components divide by this value: the guard of KAMMA_g1d_g21.py:838-842 for the arena
          // that LOGS, i.e. local arena 0 of every environment (on a shard, too, whose rew_buf[0] is not global arena 0's)
          d[27] = (T2.env0_reward_guard && fabs((double)rew) < 1e-12) ? 1e-12f : rew;
        }
      }
    }
    // [phase: projection tail (per step)]
    // tail of post_physics_step (:701-710) on the possibly reset state
    S.vfwd = forward_projection(S.q, S.v);
    // [phase: reset-event records (rare: a wave with a resetting lane)]
    // per-group record of the reset event (extras["episode"], episode statistics)
    const unsigned long long m = __ballot(do_reset);
    if (m != 0ull) {
      const OutPtrs O = kernarg_late<OutPtrs, kArgB + (int)offsetof(DevBuffers, obs)>(kp);
      group_record<LPA>(O.partials, O.stamps, O.stat_acc, arena / kGroup, lane, group_any<LPA>(m, lane), rc, S.sim_steps, true);
      // any reset_idx call re-snapshots `fraction` of every arena (KAMMA.py:498): tell level_command_kernel
      // and zeroes old_torques of every arena (KAMMA_eval.py:470): tell the next step. Two slots by the parity of
      // the step count: a wave of the NEXT parity's launch never reads the slot this launch writes
      if ((T2.cmd_sampler == OIS_CMD_LEVEL || T2.eval_sums) && lane == 0) {
        const AuxPtrs X = kernarg_late<AuxPtrs, kArgB + (int)offsetof(DevBuffers, rp_states)>(kp);
        X.reset_stamp[S.sim_steps & 1] = S.sim_steps;
      }
    }
  }
  // [phase: epilogue: state rows out (per launch)]
  if (valid) {
    // the pointers of the stores come from the kernarg segment now, not from kernel entry through the substep loop
    const RowPtrs Rp = kernarg_late<RowPtrs, kArgB + (int)offsetof(DevBuffers, pose)>(kp);
    const WheelPtrs Wp = kernarg_late<WheelPtrs, kArgB + (int)offsetof(DevBuffers, wheel_vel)>(kp);
    store_arena<LPA>(eval_sums, Rp, Wp, arena, h, S, cmd_dirty);
    if (reward_sum != nullptr && h == 0) reward_sum[arena] = rsum;
    if constexpr (TER) {
      if (B.ter.curriculum) {
        Wp.env_origins[arena] = make_float4(tor[0], tor[1], tor[2], 0.0f);
        Wp.terrain_idx[arena] = make_int4(tlevel, ttype, 0, 0);
      }
    }
  }
}

// Kernel shells around the same body. The plane kernel exists twice, and since round 4 the two differ in ONE thing,
// how many waves the hardware may put on a SIMD (the body needs 224-238 registers in both, no scratch, no spill space):
//  * step_kernel     — claims a register block > 256 of the SIMD's 512 (one_wave_per_simd below): one wave per SIMD,
//    guaranteed. Launched while the arenas give at most one wave per SIMD anyway (N <= 65 536 on the 1024 SIMDs of an
//    MI355X; the lane-pair kernel: N <= 32 768).
//  * step_kernel_w2  — no claim: two waves fit a SIMD. Launched above that size (OIS_STEP_WAVES=1|2 overrides the rule).
// Why the claim: a launch of 1024 waves that MAY double up does — not back to back, but behind the policy's kernels, whose
// last waves are still draining when the launch starts: the dispatcher fills the CUs that are free with two waves per SIMD
// and the launch takes 52 us instead of 36 (bench.py's actor loop, same binary, profiles/r4_ab_occupancy.txt). Rounds 1-3
// had this by accident (264 registers) and had measured the same effect on the then-256-register w2 shell as "cold scratch
// behind the GEMMs" (59 vs 40 us, profiles/r1_waves_ab.txt). The terrain kernels need 258-260 registers by themselves.
//  * FLAT (the packed, 4-iteration shells only): every arena stands on the default plane n = (0, 0, 1) — no
//    ois_randomize_params slopes since ois_create, no terrain: the host's knowledge (ois_env::flat) — so the
//    substeps take n_b and n.p from R and p directly (-12 instructions per substep, ois_physics_pk.h).
//  * FOLD (the packed, 4-iteration shells only): the continuation launches of a rollout under the per-level command
//    curriculum, whose prologue re-draws the commands that follow the previous step (level_command_kernel's work).
// Claims (without executing anything) an accumulation register high enough that the kernel's register block is > 256 of
// the SIMD's 512: the hardware then cannot put a second wave of this kernel on a SIMD.
template <int LPA>
OIS_DEV void one_wave_per_simd() {
  if (LPA == 1) asm volatile("" ::: "a63");   // 228 + 64
  else asm volatile("" ::: "a95");            // the lane-pair body needs 176: + 96
}

template <int LPA, bool PK, int NIT, bool FLAT = false, bool FOLD = false>
__global__ __launch_bounds__(kBlock) void step_kernel(DevConsts K, DevBuffers B, const float* __restrict__ actions,
                                                      float* __restrict__ reward_sum, int N, int ksteps, int policy,
                                                      int rp_head) {
#ifndef OIS_NO_OCCUPANCY_CLAIM
  one_wave_per_simd<LPA>();
#endif
  step_body<LPA, PK, false, NIT, FLAT, FOLD>(K.phys, B, actions, reward_sum, N, ksteps, policy, rp_head);
}
template <bool PK, int NIT, bool FLAT = false, bool FOLD = false>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(2, 2)))
void step_kernel_w2(DevConsts K, DevBuffers B, const float* __restrict__ actions, float* __restrict__ reward_sum, int N,
                    int ksteps, int policy, int rp_head) {
  step_body<1, PK, false, NIT, FLAT, FOLD>(K.phys, B, actions, reward_sum, N, ksteps, policy, rp_head);
}
// TSTRIDE: ois_terrain.footprint_stride (1: every substep samples its footprints; 2: every second one, SPEC 3.8)
template <bool PK, int NIT, bool FOLD = false, int TSTRIDE = 1>
__global__ __launch_bounds__(kBlock) void step_kernel_terrain(DevConsts K, DevBuffers B, const float* __restrict__ actions,
                                                              float* __restrict__ reward_sum, int N, int ksteps, int policy,
                                                              int rp_head) {
  step_body<1, PK, true, NIT, false, FOLD, TSTRIDE>(K.phys, B, actions, reward_sum, N, ksteps, policy, rp_head);
}

// ------------------------------------------------------------------ get_heights -------
// KAMMA.py:872-893. A workgroup takes kScanArenas consecutive arenas: their per-arena part (base-link
// position in the field's frame, the normalised yaw quaternion of quat_apply_yaw :895-899) is computed once by
// one lane per arena and parked in LDS beside the table of the 140 lattice points; then the workgroup's
// 140 * kScanArenas points are dealt to the lanes in output order, so the rows of measured_heights are written
// coalesced (non-temporal: 37 MB nobody in this kernel reads again) and a wave's lanes read the same one or two
// LDS records. The kernel is bound by the 128-byte lines its gathers pull through L2 (neither fewer instructions
// nor more gathers in flight moved it), so it reads the min-field in its 8-bit form: tiles of
// 16 x 8 cells, one int16 base per tile (in blocks of 8 x 8 tiles), the 16-bit tiles for a tile marked wide.
// Same float operations per point as the oracle's orc_get_heights and the same int16 under every point:
// bit-identical to tests/golden/refmath_heights.npz.
OIS_DEV unsigned base_index(unsigned trow, unsigned tcol, unsigned tpr) {  // 8 x 8-tile blocks of the base table
  return ((((trow >> 3) * ((tpr + 7u) >> 3) + (tcol >> 3)) << 6) | ((trow & 7u) << 3) | (tcol & 7u));
}
constexpr int kScanArenas = 32;
constexpr int kScanUnroll = 3;  // gathers a lane keeps in flight
__global__ __launch_bounds__(256) void height_scan_kernel(DevConsts K, DevBuffers B, int N, float* __restrict__ out) {
  __shared__ float4 rec[kScanArenas];        // pwx, pwy, qw, qz
  __shared__ float2 tab[OIS_HEIGHT_POINTS];  // body-frame lattice point of scan point pt (KAMMA.py:278-292)
  const int arena0 = blockIdx.x * kScanArenas;
  const int t = threadIdx.x;
  if (t < kScanArenas && arena0 + t < N) {
    const int i = arena0 + t;
    const float4 po = B.pose[i], qu = B.quat[i], eo = B.env_origins[i];
    const float q[4] = {qu.x, qu.y, qu.z, qu.w};
    float R[9];
    rotmat(q, R);
    // base-link position in the world = arena-local COM - R com + tile origin
    const float pwx = __builtin_fmaf(-R[2], K.task.com_height, po.x) + eo.x;
    const float pwy = __builtin_fmaf(-R[5], K.task.com_height, po.y) + eo.y;
    float nrm = sqrtf(qu.x * qu.x + qu.w * qu.w);  // quat_apply_yaw: x, y zeroed, then normalize (:895-899)
    nrm = fmaxf(nrm, 1e-9f);
    rec[t] = make_float4(pwx, pwy, qu.x / nrm, qu.w / nrm);
  }
  if (t >= 64 && t < 64 + OIS_HEIGHT_POINTS) {
    const int pt = t - 64;
    const int a = pt / 10, b = pt - a * 10;
    const int gx = a < 7 ? a - 8 : a - 5;   // -8..-2, 2..8   (KAMMA.py:283-285)
    const int gy = b < 5 ? b - 5 : b - 4;   // -5..-1, 1..5   (KAMMA.py:280-282)
    tab[pt] = make_float2(0.1f * (float)gx, 0.1f * (float)gy);
  }
  __syncthreads();
  const int narenas = min(kScanArenas, N - arena0);
  const int npts = narenas * OIS_HEIGHT_POINTS;
  float* o = out + (size_t)arena0 * OIS_HEIGHT_POINTS;
  const float hs = B.ter.hs, border = B.ter.border, vs = B.ter.vs;
  const float xmax = (float)(B.ter.rows - 2), ymax = (float)(B.ter.cols - 2);
  const unsigned tpr = (unsigned)B.ter.tiles_per_row;
  const char* __restrict__ m8 = reinterpret_cast<const char*>(B.ter.m8);
  const char* __restrict__ mb = reinterpret_cast<const char*>(B.ter.mb);
  const char* __restrict__ m16 = reinterpret_cast<const char*>(B.ter.m);
  // point p of the workgroup = arena la, scan point pt; a lane's next point is p + 256 = one arena + 116 points on
  int la = t >= OIS_HEIGHT_POINTS ? 1 : 0, pt = t - la * OIS_HEIGHT_POINTS;
  struct Cell { unsigned px, py; };
  auto cell = [&]() -> Cell {
    const float4 r = rec[la & (kScanArenas - 1)];  // lanes past npts read some record: their cell is clamped, never stored
    const float2 bb = tab[pt];
    const float qw = r.z, qz = r.w;
    const float bx = bb.x, by = bb.y;
    const float tx = -(qz * by) * 2.0f, ty = (qz * bx) * 2.0f;
    const float cx = -(qz * ty), cy = qz * tx;
    float X = (bx + qw * tx) + cx, Y = (by + qw * ty) + cy;
    X += r.x; Y += r.y;                       // :878
    X += border; Y += border;                 // :882
    float fx = X / hs, fy = Y / hs;           // :883
    fx = fminf(fmaxf(fx, 0.0f), xmax);        // :886-887
    fy = fminf(fmaxf(fy, 0.0f), ymax);
    pt += 256 - OIS_HEIGHT_POINTS; la += 1;
    if (pt >= OIS_HEIGHT_POINTS) { pt -= OIS_HEIGHT_POINTS; la += 1; }
    return {(unsigned)(int)fx, (unsigned)(int)fy};
  };
  // :889-893: the minimum of the two samples is in the min-field; 8-bit tile + base, the 16-bit field for a wide tile
  // (32-bit byte offsets: ois_terrain_attach refuses a field of 2^31 bytes)
  auto tile_of = [&](Cell c) -> unsigned { return (c.px >> 4) * tpr + (c.py >> 3); };
  auto load_base = [&](Cell c) -> int { return *reinterpret_cast<const int16_t*>(mb + (base_index(c.px >> 4, c.py >> 3, tpr) << 1)); };
  auto load_off = [&](unsigned tl, Cell c) -> int {
    return *reinterpret_cast<const uint8_t*>(m8 + ((tl << 7) + (((c.px & 15u) << 3) | (c.py & 7u))));
  };
  auto finish = [&](Cell c, int base, int off) -> float {
    int hq = base + off;
    if (__builtin_expect(base == kWideTile, 0))
      hq = *reinterpret_cast<const int16_t*>(m16 + ((((c.px >> 3) * tpr + (c.py >> 3)) << 7) + (((c.px & 7u) << 4) | ((c.py & 7u) << 1))));
    return (float)hq * vs;
  };
  int p = t;
  const int chunks = npts / (256 * kScanUnroll);  // whole chunks: every lane has a point in each of the rows
  for (int c = 0; c < chunks; ++c, p += 256 * kScanUnroll) {
    Cell ce[kScanUnroll];
    int base[kScanUnroll], off[kScanUnroll];
#pragma unroll
    for (int u = 0; u < kScanUnroll; ++u) ce[u] = cell();
#pragma unroll
    for (int u = 0; u < kScanUnroll; ++u) { const unsigned tl = tile_of(ce[u]); base[u] = load_base(ce[u]); off[u] = load_off(tl, ce[u]); }
#pragma unroll
    for (int u = 0; u < kScanUnroll; ++u) __builtin_nontemporal_store(finish(ce[u], base[u], off[u]), o + p + 256 * u);
  }
  for (; p < npts; p += 256) {
    const Cell c1 = cell();
    const unsigned tl = tile_of(c1);
    __builtin_nontemporal_store(finish(c1, load_base(c1), load_off(tl, c1)), o + p);
  }
}

// m[x][y] = min(h[x][y], h[x+1][y+1]) (last row / column: h itself; never read, see ois.h)
OIS_DEV int16_t minfield_value(const int16_t* __restrict__ h, int x, int y, int rows, int cols) {
  const int16_t a = h[(size_t)x * cols + y];
  int16_t b = a;
  if (x + 1 < rows && y + 1 < cols) b = h[(size_t)(x + 1) * cols + y + 1];
  return a < b ? a : b;
}
__global__ __launch_bounds__(256) void minfield_kernel(const int16_t* __restrict__ h, int16_t* __restrict__ m, int rows, int cols) {
  const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= (size_t)rows * cols) return;
  const int x = (int)(t / cols), y = (int)(t % cols);
  const size_t o = ((size_t)((x >> 3) * ((cols + 7) >> 3) + (y >> 3)) << 6) + (size_t)(((x & 7) << 3) | (y & 7));  // 8 x 8 tiles
  m[o] = minfield_value(h, x, y, rows, cols);
}
// the same field for the height scan: tiles of 16 x 8 cells as 8-bit offsets from the tile's minimum (one 128-byte
// line holds twice the cells: the scan is bound by the lines it pulls through L2). A tile whose
// heights span more than 254 units gets the base kWideTile and is read from the 16-bit field. One thread per tile.
__global__ __launch_bounds__(64) void minfield8_kernel(const int16_t* __restrict__ h, uint8_t* __restrict__ m8, int16_t* __restrict__ mb,
                                                        int rows, int cols) {
  const int tpr = (cols + 7) >> 3, ntiles = ((rows + 15) >> 4) * tpr;
  const int tl = blockIdx.x * 64 + threadIdx.x;
  if (tl >= ntiles) return;
  const int x0 = (tl / tpr) << 4, y0 = (tl % tpr) << 3;
  int lo = 32767, hi = -32768;
  for (int i = 0; i < 128; ++i) {
    const int x = x0 + (i >> 3), y = y0 + (i & 7);
    if (x < rows && y < cols) { const int v = minfield_value(h, x, y, rows, cols); lo = v < lo ? v : lo; hi = v > hi ? v : hi; }
  }
  const bool wide = hi - lo > 254 || lo == kWideTile;
  mb[base_index(tl / tpr, tl % tpr, tpr)] = (int16_t)(wide ? kWideTile : lo);
  for (int i = 0; i < 128; ++i) {
    const int x = x0 + (i >> 3), y = y0 + (i & 7);
    const int v = (!wide && x < rows && y < cols) ? minfield_value(h, x, y, rows, cols) - lo : 0;
    m8[((size_t)tl << 7) + i] = (uint8_t)v;
  }
}

// ------------------------------------------------------------------ post_reset --------
// KAMMA.py:366-414: zero the task buffers, reset_idx(all arenas). One lane per arena.
__global__ __launch_bounds__(kBlock) void reset_kernel(DevConsts K, DevBuffers B, int N) {
  const int arena = blockIdx.x * kBlock + threadIdx.x;
  const int lane = threadIdx.x & 63;
  const bool valid = arena < N;
  const int la = valid ? arena : N - 1;
  ArenaT<1> S;
  for (int k = 0; k < 3; ++k) { S.p[k] = 0.0f; S.v[k] = 0.0f; S.w[k] = 0.0f; S.n[k] = 0.0f; }
  for (int k = 0; k < 4; ++k) {
    S.q[k] = 0.0f; S.cmd[k] = 0.0f; S.jv[k] = 0.0f; S.ln[k] = 0.0f; S.l1[k] = 0.0f; S.l2[k] = 0.0f; S.wt[k] = 0.0f;
    S.sa[k] = 0.0f; S.sb[k] = 0.0f; S.sc[k] = 0.0f;
  }
  S.vfwd = S.prev_lin = S.prev_ang = S.mu = 0.0f;
  S.ep_buf = S.sim_steps = S.ep_count = 0;
  // gamma_assist1/2 hold their constructor values here (KAMMA.py:192-193: ones)
  const float g1 = K.task.gamma1_mode == OIS_GAMMA_CONST ? K.task.gamma1_const : 1.0f;
  const float g2 = K.task.gamma2_mode == OIS_GAMMA_CONST ? K.task.gamma2_const : 1.0f;
  ResetContrib rc;
  if (B.ter.on) reset_arena<1, false>(K.task, K.reset, K.task.arena_offset + la, 0, g1, g2, S, rc);
  else {  // on the arena's own plane (params are owned by ois_create / ois_randomize_params)
    const float4 pr = B.params[la];
    S.n[0] = pr.y; S.n[1] = pr.z; S.n[2] = pr.w;
    reset_arena<1, true>(K.task, K.reset, K.task.arena_offset + la, 0, g1, g2, S, rc);
  }
  if (!valid) contrib_identity(rc);
  if ((lane & (kGroup - 1)) == 0 && valid) {  // clear the finished-episode accumulators of this group
    double* Acc = B.stat_acc + (size_t)(arena / kGroup) * 8;
    for (int k = 0; k < 4; ++k) { Acc[k] = 0.0; Acc[4 + k] = -__builtin_inf(); }
  }
  const unsigned long long m = __ballot(valid);
  group_record<1>(B.partials, B.stamps, B.stat_acc, arena / kGroup, lane, group_any<1>(m, lane), rc, 0, false);
  if (valid) {
    // params are owned by ois_create / ois_randomize_params: keep them
    B.pose[arena] = make_float4(S.p[0], S.p[1], S.p[2], S.vfwd);
    B.quat[arena] = make_float4(S.q[0], S.q[1], S.q[2], S.q[3]);
    B.linvel[arena] = make_float4(S.v[0], S.v[1], S.v[2], S.prev_lin);
    B.angvel[arena] = make_float4(S.w[0], S.w[1], S.w[2], S.prev_ang);
    B.commands[arena] = make_float4(S.cmd[0], S.cmd[1], S.cmd[2], S.cmd[3]);
    B.counters[arena] = make_int4(S.ep_buf, S.sim_steps, S.ep_count, 0);
    B.wheel_vel[arena] = make_float4(S.jv[0], S.jv[1], S.jv[2], S.jv[3]);
    const float4 z = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    B.lam_n[arena] = make_uint2(0u, 0u); B.lam_l[arena] = make_uint2(0u, 0u); B.lam_t[arena] = make_uint2(0u, 0u);
    B.sums_a[arena] = z; B.sums_b[arena] = z; B.sums_c[arena] = z; B.wheel_torqs[arena] = z;
    for (int k = 0; k < 6; ++k) B.obs[(size_t)arena * 6 + k] = 0.0f;
    B.rew[arena] = 0.0f; B.done[arena] = 0; B.timeout[arena] = 0;
  }
  if (arena < OIS_ENV0_DIAG_LEN) { B.env0_diag[arena] = 0.0f; B.episode_info[arena] = 0.0f; }
  if (valid) B.cmd_fraction[arena] = 0.0f;  // KAMMA.py:498 at post_reset: the sums are zero
  if (arena == 0) { B.reset_stamp[0] = -1; B.reset_stamp[1] = -1; }
}

// Tail of post_physics_step with terrain.curriculum (KAMMA.py:719-725): the command of EVERY arena is
// re-drawn from its terrain level's generator after every step; `fraction` (KAMMA.py:498) is
// re-snapshot for all arenas when any arena reset in the step just done. Its own launch because that
// "any arena" is only known once every wave of the step kernel has finished.
__global__ __launch_bounds__(256) void level_command_kernel(DevConsts K, DevBuffers B, int N) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= N) return;
  const int sim = B.counters[i].y;  // == episode_step_counter: policy steps since post_reset
  float fr = B.cmd_fraction[i];
  if (B.reset_stamp[sim & 1] == sim) {
    fr = B.sums_a[i].x / K.reset.cmd_thr_high;
    B.cmd_fraction[i] = fr;
  }
  float u[4];
  draw4(K.task, K.task.arena_offset + i, (uint32_t)sim, PURPOSE_CMD, u);
  const int level = B.ter.on ? B.terrain_idx[i].x : 0;  // zeros without a terrain (KAMMA.py:223)
  float cx, cw;
  level_command(level, fr, sim, normal_det(u[0], u[1]), K.reset.cmd_dt, K.reset.cmd_episode_s, cx, cw);
  float4 c = B.commands[i];
  c.x = cx; c.z = cw;
  B.commands[i] = c;
}

// push_robots (KAMMA.py:729-733): base_velocities[:, 0:2] = torch_rand_float(-1.0, 1.0, (num_envs, 2))
__global__ __launch_bounds__(256) void push_robots_kernel(DevConsts K, DevBuffers B, int N) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= N) return;
  float u[4];
  draw4(K.task, K.task.arena_offset + i, (uint32_t)B.counters[i].y, PURPOSE_PUSH, u);
  float4 v = B.linvel[i];
  v.x = rand_float(-1.0f, 1.0f, u[0]);
  v.y = rand_float(-1.0f, 1.0f, u[1]);
  B.linvel[i] = v;
}

__global__ void init_params_kernel(DevBuffers B, int N, float mu) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < N) B.params[i] = make_float4(mu, 0.0f, 0.0f, 1.0f);
}

// per-arena domain randomisation (BASELINE.json config 4)
__global__ void randomize_params_kernel(DevConsts K, DevBuffers B, int N, float mu_lo, float mu_hi,
                                        const float* __restrict__ slopes, int n_slopes) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  float u[4];
  draw4(K.task, K.task.arena_offset + i, 0u, PURPOSE_PARAMS, u);
  float nx = 0.0f, ny = 0.0f, nz = 1.0f;
  if (n_slopes > 0) {
    int idx = (int)(u[1] * (float)n_slopes);
    if (idx >= n_slopes) idx = n_slopes - 1;
    const float sl = slopes[idx];
    float s, c;
    sincos_det(3.14159274101257324f * u[2], s, c);
    const float inv = 1.0f / sqrtf(__builtin_fmaf(sl, sl, 1.0f));
    nx = -(sl * c) * inv; ny = -(sl * s) * inv; nz = inv;
  }
  B.params[i] = make_float4(rand_float(mu_lo, mu_hi, u[0]), nx, ny, nz);
}

// extras["episode"] (KAMMA.py:480-488): deterministic two-level sum of the per-group records of
// the most recent reset event (docs/SPEC.md §2.10). Level 1: one wave per chunk of kChunkGroups
// consecutive groups sums, in group order, the records carrying the chunk's newest stamp;
// level 2: the chunk partials whose newest stamp is the global newest are summed in chunk order.
// Since round 4 every load of a level is in flight before the first add — the sums are chains of
// 64 dependent adds, not of 64 dependent round trips to L2 (two launches of serial loads took 18.2 us at 65 536
// arenas; the reference trainer reads infos after every step). The finishing wave then leaves the 32 values + the 32
// env-0 diagnostics of the step in the library's pinned host block and publishes a sequence number behind them
// (ois_logs_*): the host waits on a word in its own memory instead of draining the stream through a blocking copy.
// Two forms: logs_block_kernel (one workgroup, one launch: the default, below) and, for more than kBlockChunks chunks or
// a terrain whose levels move (its level mean is a sum over all arenas), logs_level1_kernel (one wave per chunk) +
// logs_level2_kernel: two launches, because the end of a kernel is the cheapest grid-wide barrier this part has — a
// "wave with the last ticket does level 2" version of the same took 14.5 us per launch.
constexpr int kChunkGroups = 64;  // 1024 arenas per chunk

OIS_DEV int wave_max_i(int x) {
#pragma unroll
  for (int w = 32; w >= 1; w >>= 1) { const int y = __shfl_xor(x, w, 64); x = y > x ? y : x; }
  return x;
}
// one level of the record sum for slot `t` (0..31; called by all 64 lanes, the upper half shadows the lower): the
// up-to-64 rows at `rows`, stride 32 floats, of which row i counts when lane i's `stamp` equals `newest`; slots
// 24 / 25 (gamma1 / gamma2, identical in every record of one event) are max-reduced. All loads first, then the adds
// in row order.
OIS_DEV float record_level(const float* __restrict__ rows, int nrows, int stamp, int newest, int t, float a) {
  const bool is_gamma = t == 24 || t == 25;
  float x[kChunkGroups];
#pragma unroll
  for (int i = 0; i < kChunkGroups; ++i) x[i] = rows[(size_t)(i < nrows ? i : nrows - 1) * 32 + t];
#pragma unroll
  for (int i = 0; i < kChunkGroups; ++i) {
    const bool match = __builtin_amdgcn_readlane(stamp, i) == newest;  // lanes >= nrows carry INT_MIN: never the newest
    const float y = is_gamma ? fmaxf(a, x[i]) : a + x[i];
    a = match ? y : a;
  }
  return a;
}

struct LogsArgs {
  float* part;          // [nchunks][32] level-1 partials
  int* cmax;            // [nchunks] newest stamp of each chunk
  long long* lvl;       // [nchunks] terrain-level sums
  float level_mean;     // torch.mean(terrain_levels) when the host knows it (a terrain without curriculum: set at attach); else < 0
  float* host_block;    // pinned host memory: [32] episode info | [32] env-0 diagnostics | uint64 sequence; or null
  unsigned long long seq;
};

__global__ __launch_bounds__(64) void logs_level1_kernel(DevBuffers B, int ngroups, int N, LogsArgs A) {
  const int c = blockIdx.x, t = threadIdx.x;
  constexpr int kNever = -2147483647 - 1;
  {
    const int g0 = c * kChunkGroups, n = min(kChunkGroups, ngroups - g0);
    const int st = t < n ? B.stamps[g0 + t] : kNever;
    const int m = wave_max_i(st);
    const bool is_gamma = (t & 31) == 24 || (t & 31) == 25;
    const float p = record_level(B.partials + (size_t)g0 * 32, n, st, m, t & 31, is_gamma ? -__builtin_inff() : 0.0f);
    if (t < 32) A.part[(size_t)c * 32 + t] = p;
    if (t == 0) A.cmax[c] = m;
    // :488 torch.mean(terrain_levels.float()) over ALL arenas: integer sums, exact and order-free
    if (B.ter.on && A.level_mean < 0.0f) {
      long long ls = 0;
      const int a1 = min((g0 + n) * kGroup, N);
      for (int i = g0 * kGroup + t; i < a1; i += 64) ls += B.terrain_idx[i].x;
      for (int w = 32; w >= 1; w >>= 1) ls += __shfl_xor(ls, w, 64);
      if (t == 0) A.lvl[c] = ls;
    }
  }
}

__global__ __launch_bounds__(64) void logs_level2_kernel(DevConsts K, DevBuffers B, int nchunks, int N, LogsArgs A) {
  const int t = threadIdx.x;
  constexpr int kNever = -2147483647 - 1;
  int sm = kNever;
  for (int i = t; i < nchunks; i += 64) sm = max(sm, A.cmax[i]);
  const int smax = wave_max_i(sm);
  const bool is_gamma = (t & 31) == 24 || (t & 31) == 25;
  float a = is_gamma ? -__builtin_inff() : 0.0f;
  for (int c0 = 0; c0 < nchunks; c0 += kChunkGroups) {
    const int n = min(kChunkGroups, nchunks - c0);
    const int cm = t < n ? A.cmax[c0 + t] : kNever;
    a = record_level(A.part + (size_t)c0 * 32, n, cm, smax, t & 31, a);
  }
  long long tot = 0;
  if (B.ter.on && A.level_mean < 0.0f) {
    for (int i = t; i < nchunks; i += 64) tot += A.lvl[i];
    for (int w = 32; w >= 1; w >>= 1) tot += __shfl_xor(tot, w, 64);
  }
  // record slot -> extras slot (include/ois.h ois_episode_info_slot): lane s < 32 produces I[s]
  const int s = t & 31;
  const int src = (s >= OIS_EI_R1 && s <= OIS_EI_OBSERVED_OVER_FINAL) ? s + 3
                : (s >= OIS_EI_SMOOTHNESS && s <= OIS_EI_DESIRED_VELOCITY) ? s
                : s == OIS_EI_GAMMA1 ? 24 : s == OIS_EI_GAMMA2 ? 25 : 0;
  const float cnt = __shfl(a, 0, 64), v = __shfl(a, src, 64);
  float out = 0.0f;
  if ((s >= OIS_EI_R1 && s <= OIS_EI_OBSERVED_OVER_FINAL) || (s >= OIS_EI_SMOOTHNESS && s <= OIS_EI_DESIRED_VELOCITY))
    out = (v / cnt) / K.reset.episode_length_s;                            // :482-484
  else if (s == OIS_EI_COUNT || s == OIS_EI_GAMMA1 || s == OIS_EI_GAMMA2) out = v;  // :486, :487
  else if (s == OIS_EI_TERRAIN_LEVEL) out = A.level_mean >= 0.0f ? A.level_mean : (float)tot / (float)N;   // :488 (zeros without a terrain, :223)
  else if (s == OIS_EI_STEP) out = (float)smax;
  if (t < 32) B.episode_info[t] = out;
  else out = B.env0_diag[t - 32];
  if (A.host_block != nullptr) {
    A.host_block[t] = out;
    __threadfence_system();
    if (t == 0) __hip_atomic_store(reinterpret_cast<unsigned long long*>(A.host_block + 64), A.seq, __ATOMIC_RELEASE,
                                   __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// The same in ONE WORKGROUP and one launch, for environments of up to kBlockChunks chunks (131 072 arenas) on the plane or on a
// terrain whose levels cannot move — the case a trainer's loop is in after every step. The 8 XCDs of the part do not share an
// L2: a single-launch version with one wave per chunk and a "last ticket" (an agent-scope release per block = a write-back of
// the XCD's dirty L2, right behind a step kernel that stored 26 MB; 64 same-address atomics that serialise at the memory side)
// took 14.5 us per launch in the trainer's loop (rocprofv3), for 0.5 MB of reads. One workgroup needs no cross-XCD hop at all:
// 16 waves take the chunks two at a time (lanes 0-31 one chunk, lanes 32-63 the next: the 32 record slots), level 2 goes
// through 16 KB of LDS in chunk order. And it reads only what changed:
//  * the newest stamp first (16 KB). If it is the one the published record already carries (no arena reset in this step —
//    98 steps of 99 while the episodes run in step), the record is re-published as it stands;
//  * otherwise only the rows carrying the newest stamp are loaded (a few groups after a fall, all of them after a timeout).
// Same two-level order, same values, bit for bit (tests/test_gpu_parity.py::test_one_launch_logging_reductions_against_the_oracle).
constexpr int kBlockChunks = 128;
constexpr int kLogsThreads = 1024;
__global__ __launch_bounds__(kLogsThreads) void logs_block_kernel(DevConsts K, DevBuffers B, int ngroups, int nchunks, int N, LogsArgs A,
                                                                 int force) {
  __shared__ float s_part[kBlockChunks][32];
  __shared__ int s_cmax[kBlockChunks];
  __shared__ int s_red[kLogsThreads / 64];
  constexpr int kNever = -2147483647 - 1;
  const int t = threadIdx.x, lane = t & 63, w = t >> 6, h = lane >> 5, slot = lane & 31;
  int sm = kNever;
  for (int g = t; g < ngroups; g += kLogsThreads) sm = max(sm, B.stamps[g]);
  sm = wave_max_i(sm);
  if (lane == 0) s_red[w] = sm;
  __syncthreads();
  int smax = s_red[0];
#pragma unroll
  for (int k = 1; k < kLogsThreads / 64; ++k) smax = max(smax, s_red[k]);
  const bool fresh = force != 0 || B.episode_info[OIS_EI_STEP] != (float)smax;  // block-uniform
  const bool is_gamma = slot == 24 || slot == 25;
  if (fresh) {
    for (int c0 = 2 * w; c0 < nchunks; c0 += 2 * (kLogsThreads / 64)) {
      const int ga = c0 * kChunkGroups, na = min(kChunkGroups, ngroups - ga);
      const int gb = ga + kChunkGroups, nb = c0 + 1 < nchunks ? min(kChunkGroups, ngroups - gb) : 0;
      const int st_a = lane < na ? B.stamps[ga + lane] : kNever;
      const int st_b = lane < nb ? B.stamps[gb + lane] : kNever;
      const int ma = wave_max_i(st_a), mb = wave_max_i(st_b);
      const float* rows = B.partials + (size_t)(h ? gb : ga) * 32 + slot;
      float x[kChunkGroups];
#pragma unroll
      for (int i = 0; i < kChunkGroups; ++i) {  // the loads of the rows that count, all in flight before the first add
        const int sa = __builtin_amdgcn_readlane(st_a, i), sb = __builtin_amdgcn_readlane(st_b, i);
        x[i] = 0.0f;
        if ((h ? sb : sa) == smax) x[i] = rows[(size_t)i * 32];
      }
      float a = is_gamma ? -__builtin_inff() : 0.0f;
#pragma unroll
      for (int i = 0; i < kChunkGroups; ++i) {  // row order
        const int sa = __builtin_amdgcn_readlane(st_a, i), sb = __builtin_amdgcn_readlane(st_b, i);
        const float y = is_gamma ? fmaxf(a, x[i]) : a + x[i];
        a = (h ? sb : sa) == smax ? y : a;
      }
      const int c = c0 + h;
      if (c < nchunks) {
        s_part[c][slot] = a;
        if (slot == 0) s_cmax[c] = h ? mb : ma;
      }
    }
  }
  __syncthreads();
  if (w != 0) return;
  float out;
  if (fresh) {
    float a = is_gamma ? -__builtin_inff() : 0.0f;
    for (int c = 0; c < nchunks; ++c) {  // chunk order
      const float x = s_part[c][slot];
      const float y = is_gamma ? fmaxf(a, x) : a + x;
      a = s_cmax[c] == smax ? y : a;
    }
    const int s = slot;
    const int src = (s >= OIS_EI_R1 && s <= OIS_EI_OBSERVED_OVER_FINAL) ? s + 3
                  : (s >= OIS_EI_SMOOTHNESS && s <= OIS_EI_DESIRED_VELOCITY) ? s
                  : s == OIS_EI_GAMMA1 ? 24 : s == OIS_EI_GAMMA2 ? 25 : 0;
    const float cnt = __shfl(a, 0, 64), v = __shfl(a, src, 64);
    out = 0.0f;
    if ((s >= OIS_EI_R1 && s <= OIS_EI_OBSERVED_OVER_FINAL) || (s >= OIS_EI_SMOOTHNESS && s <= OIS_EI_DESIRED_VELOCITY))
      out = (v / cnt) / K.reset.episode_length_s;                            // :482-484
    else if (s == OIS_EI_COUNT || s == OIS_EI_GAMMA1 || s == OIS_EI_GAMMA2) out = v;  // :486, :487
    else if (s == OIS_EI_TERRAIN_LEVEL) out = A.level_mean >= 0.0f ? A.level_mean : 0.0f;  // :488; zeros without a terrain (:223)
    else if (s == OIS_EI_STEP) out = (float)smax;
    if (lane < 32) B.episode_info[lane] = out;
  } else {
    out = B.episode_info[slot];
  }
  if (lane >= 32) out = B.env0_diag[lane - 32];
  if (A.host_block != nullptr) {
    A.host_block[lane] = out;
    __threadfence_system();
    if (lane == 0) __hip_atomic_store(reinterpret_cast<unsigned long long*>(A.host_block + 64), A.seq, __ATOMIC_RELEASE,
                                      __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// finished-episode statistics: the same two-level order over the per-group accumulators, in double
// (sum of squared returns of millions of episodes: the variance is a difference of two large numbers); two launches
// (level 1 per chunk, level 2 in one wave), the loads of a level in flight together
OIS_DEV double stat_level(const double* __restrict__ rows, int nrows, int t, double a) {
  double x[kChunkGroups];
#pragma unroll
  for (int i = 0; i < kChunkGroups; ++i) x[i] = rows[(size_t)(i < nrows ? i : nrows - 1) * 8 + t];
#pragma unroll
  for (int i = 0; i < kChunkGroups; ++i) {
    const double y = t < 4 ? a + x[i] : fmax(a, x[i]);
    a = i < nrows ? y : a;
  }
  return a;
}
__global__ __launch_bounds__(64) void stats_level1_kernel(DevBuffers B, int ngroups, double* __restrict__ part8, int reset_after) {
  const int c = blockIdx.x, t = threadIdx.x;
  const int g0 = c * kChunkGroups, n = min(kChunkGroups, ngroups - g0);
  if (t >= 8) return;
  double* rows = B.stat_acc + (size_t)g0 * 8;
  part8[(size_t)c * 8 + t] = stat_level(rows, n, t, t < 4 ? 0.0 : -__builtin_inf());
  if (reset_after) for (int i = 0; i < n; ++i) rows[(size_t)i * 8 + t] = t < 4 ? 0.0 : -__builtin_inf();
}
__global__ __launch_bounds__(64) void stats_level2_kernel(int nchunks, const double* __restrict__ part8, double* __restrict__ out8) {
  const int t = threadIdx.x;
  if (t >= 8) return;
  double a = t < 4 ? 0.0 : -__builtin_inf();
  for (int c0 = 0; c0 < nchunks; c0 += kChunkGroups) a = stat_level(part8 + (size_t)c0 * 8, min(kChunkGroups, nchunks - c0), t, a);
  out8[t] = a;
}

// VecEnv.step hands the caller FRESH tensors (the wrappers clone obs / rew / reset_buf and make a new all-false
// `truncated`): the four copies in one launch instead of three device-to-device copies and a fill
__global__ __launch_bounds__(256) void outputs_kernel(DevBuffers B, int N, float* __restrict__ obs, float* __restrict__ rew,
                                                      uint8_t* __restrict__ term, uint8_t* __restrict__ trunc) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= N) return;
  if (obs) {
    const float2* s = reinterpret_cast<const float2*>(B.obs + (size_t)i * 6);
    float2* o = reinterpret_cast<float2*>(obs + (size_t)i * 6);
    o[0] = s[0]; o[1] = s[1]; o[2] = s[2];
  }
  if (rew) rew[i] = B.rew[i];
  if (term) term[i] = B.done[i];
  if (trunc) trunc[i] = 0;
}

// ------------------------------------------------------------------ replay sampling ---
// One thread per sampled transition: 69 B gathered from 5 rows, written to 5 contiguous batches.
__global__ __launch_bounds__(256) void replay_sample_kernel(DevBuffers B, size_t rows, const int64_t* __restrict__ idx, int count,
                                                            float* __restrict__ st, float* __restrict__ ac, float* __restrict__ rw,
                                                            float* __restrict__ nx, uint8_t* __restrict__ dn) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= count) return;
  int64_t r = idx[i];
  r = r < 0 ? 0 : (r >= (int64_t)rows ? (int64_t)rows - 1 : r);
  if (st) {
    const float2* s = reinterpret_cast<const float2*>(B.rp_states + (size_t)r * 6);
    float2* o = reinterpret_cast<float2*>(st + (size_t)i * 6);
    o[0] = s[0]; o[1] = s[1]; o[2] = s[2];
  }
  if (nx) {
    const float2* s = reinterpret_cast<const float2*>(B.rp_next + (size_t)r * 6);
    float2* o = reinterpret_cast<float2*>(nx + (size_t)i * 6);
    o[0] = s[0]; o[1] = s[1]; o[2] = s[2];
  }
  if (ac) reinterpret_cast<float4*>(ac)[i] = reinterpret_cast<const float4*>(B.rp_actions)[r];
  if (rw) rw[i] = B.rp_rewards[r];
  if (dn) dn[i] = B.rp_done[r];
}

// ------------------------------------------------------------------ host side ---------
thread_local char g_err[512] = "";
int fail(int code, const char* fmt, const char* detail = "") {
  snprintf(g_err, sizeof g_err, fmt, detail);
  return code;
}
#define HIP_TRY(expr)                                                            \
  do {                                                                           \
    hipError_t e_ = (expr);                                                      \
    if (e_ != hipSuccess) {                                                      \
      snprintf(g_err, sizeof g_err, "%s failed: %s", #expr, hipGetErrorString(e_)); \
      return OIS_EHIP;                                                           \
    }                                                                            \
  } while (0)

size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

// Makes `device` current for the scope of one entry point and puts the caller's device back on the
// way out: a process that drives several GPUs (torch on cuda:0, an env on cuda:1) keeps its own
// current device across library calls.
struct DeviceGuard {
  int prev = -1;
  hipError_t err = hipSuccess;
  explicit DeviceGuard(int device) {
    err = hipGetDevice(&prev);
    if (err == hipSuccess && prev != device) err = hipSetDevice(device);
    else if (err == hipSuccess) prev = -1;  // already current: nothing to restore
  }
  ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};
#define OIS_ON_DEVICE(dev_)            \
  DeviceGuard device_guard_(dev_);     \
  HIP_TRY(device_guard_.err)

struct Layout {
  size_t off[OIS_BUF_COUNT_];
  size_t bytes[OIS_BUF_COUNT_];
  size_t partials, stamps, stat_acc, stat_out, red_part, red_cmax, red_lvl, red_part8, red_ticket, reset_stamp, total;
  int nwaves, nchunks;
};

Layout make_layout(int N) {
  Layout L;
  std::memset(&L, 0, sizeof L);
  const size_t f4 = (size_t)N * 4 * sizeof(float);
  size_t o = 0;
  auto put = [&](int id, size_t b) { L.off[id] = o; L.bytes[id] = b; o = align256(o + b); };
  // persistent state first (ois_save_state copies [0, total))
  put(OIS_BUF_POSE, f4); put(OIS_BUF_QUAT, f4); put(OIS_BUF_LINVEL, f4); put(OIS_BUF_ANGVEL, f4);
  put(OIS_BUF_WHEEL_VEL, f4); put(OIS_BUF_COMMANDS, f4); put(OIS_BUF_COUNTERS, f4);
  put(OIS_BUF_SUMS_A, f4); put(OIS_BUF_SUMS_B, f4); put(OIS_BUF_SUMS_C, f4); put(OIS_BUF_PARAMS, f4);
  put(OIS_BUF_LAMBDA_N, f4 / 2); put(OIS_BUF_LAMBDA_L, f4 / 2); put(OIS_BUF_LAMBDA_T, f4 / 2);  // 4 halves per arena
  put(OIS_BUF_WHEEL_TORQS, f4);
  put(OIS_BUF_ENV_ORIGINS, f4); put(OIS_BUF_TERRAIN_IDX, f4);
  put(OIS_BUF_CMD_FRACTION, (size_t)N * sizeof(float));
  put(OIS_BUF_OBS, (size_t)N * 6 * sizeof(float));
  put(OIS_BUF_REW, (size_t)N * sizeof(float));
  put(OIS_BUF_DONE, (size_t)N);
  put(OIS_BUF_TIMEOUT, (size_t)N);
  put(OIS_BUF_EPISODE_INFO, OIS_EPISODE_INFO_LEN * sizeof(float));
  put(OIS_BUF_ENV0_DIAG, OIS_ENV0_DIAG_LEN * sizeof(float));
  L.nwaves = (N + kGroup - 1) / kGroup;  // reduction groups of 16 arenas
  L.partials = o; o = align256(o + (size_t)L.nwaves * 32 * sizeof(float));
  L.stamps = o; o = align256(o + (size_t)L.nwaves * sizeof(int32_t));
  L.stat_acc = o; o = align256(o + (size_t)L.nwaves * 8 * sizeof(double));
  L.stat_out = o; o = align256(o + 32 * sizeof(double));  // 8 stats out (double) / 16-entry slope table scratch (float)
  L.nchunks = (L.nwaves + kChunkGroups - 1) / kChunkGroups;  // level-1 partials of the logging reductions
  L.red_part = o; o = align256(o + (size_t)L.nchunks * 32 * sizeof(float));
  L.red_cmax = o; o = align256(o + (size_t)L.nchunks * sizeof(int));
  L.red_lvl = o; o = align256(o + (size_t)L.nchunks * sizeof(long long));
  L.red_part8 = o; o = align256(o + (size_t)L.nchunks * 8 * sizeof(double));
  L.red_ticket = o; o = align256(o + 2 * sizeof(unsigned));  // reserved (the ticket words of round 4's first logs kernels); keeps the blob layout
  L.reset_stamp = o; o = align256(o + 2 * sizeof(int32_t));
  L.total = o;
  return L;
}

int validate(const ois_config* c) {
  if (!c) return fail(OIS_EINVAL, "null config");
  if (c->struct_bytes != sizeof(ois_config)) return fail(OIS_EINVAL, "ois_config.struct_bytes does not match this library");
  if (c->num_envs <= 0 || c->num_envs > (1 << 27)) return fail(OIS_EINVAL, "num_envs out of range");
  if (c->action_dim != 4 && c->action_dim != 2 && c->action_dim != 1) return fail(OIS_EINVAL, "action_dim must be 4, 2 or 1");
  if (c->solver_iterations < 1 || c->solver_iterations > 16) return fail(OIS_EINVAL, "solver_iterations out of range");
  if (c->decimation < 0 || c->extra_substeps < 0 || c->decimation + c->extra_substeps < 1 || c->decimation + c->extra_substeps > 1000)
    return fail(OIS_EINVAL, "decimation + extra_substeps out of range");
  if (!(c->sim_dt > 0.0f) || !(c->policy_dt > 0.0f)) return fail(OIS_EINVAL, "time steps must be positive");
  if (!(c->body_mass > 0.0f) || !(c->wheel_spin_inertia > 0.0f) || !(c->body_inertia[0] > 0.0f) ||
      !(c->body_inertia[1] > 0.0f) || !(c->body_inertia[2] > 0.0f) || !(c->wheel_radius > 0.0f))
    return fail(OIS_EINVAL, "mass / inertia / radius must be positive");
  if (c->mix_mode != OIS_MIX_SWITCH && c->mix_mode != OIS_MIX_BLEND) return fail(OIS_EINVAL, "bad mix_mode");
  if (c->cmd_sampler < 0 || c->cmd_sampler > OIS_CMD_LEVEL) return fail(OIS_EINVAL, "bad cmd_sampler");
  if (c->cmd_sampler == OIS_CMD_LEVEL && !(c->curriculum_threshold_high > 0.0f)) return fail(OIS_EINVAL, "curriculum_threshold_high must be positive");
  if (!(c->max_sim_steps > 0.0f)) return fail(OIS_EINVAL, "max_sim_steps must be positive");
  if (!(c->baumgarte > 0.0f && c->baumgarte < 1.0f)) return fail(OIS_EINVAL, "baumgarte must be in (0, 1)");
  if (!(c->wheel_damping >= 0.0f) || !(c->wheel_limit_damping >= 0.0f) || !(c->wheel_limit_velocity >= 0.0f))
    return fail(OIS_EINVAL, "wheel joint resistance constants must be >= 0");
  // the joint resistance is integrated explicitly, spin += dt / Is * (tau - c1 w - c2 over) (SPEC 3.3): it oscillates
  // with growing amplitude once (c1 + c2) dt / Is reaches 2
  if (!((c->wheel_damping + c->wheel_limit_damping) * c->sim_dt / c->wheel_spin_inertia < 2.0f))
    return fail(OIS_EINVAL, "(wheel_damping + wheel_limit_damping) * sim_dt / wheel_spin_inertia must be < 2 (explicit integration)");
  // the normal-row bias is min(-gap / dt, -beta gap / dt, v_dep) (SPEC 3.4): v_dep < 0 would push separating contacts
  if (!(c->max_depenetration_velocity >= 0.0f)) return fail(OIS_EINVAL, "max_depenetration_velocity must be >= 0");
  // the friction cone's half-width is mu * lambda_n (SPEC 3.5): a negative or non-finite coefficient turns the clamp inside out
  if (!(c->friction >= 0.0f) || !(c->friction < 1e6f)) return fail(OIS_EINVAL, "friction must be >= 0 (finite)");
  return OIS_OK;
}

// same derivation, same float operations as the oracle's derive() (docs/SPEC.md §3.0)
void derive(const ois_config* c, DevConsts* K) {
  std::memset(K, 0, sizeof *K);
  K->phys.dt = c->sim_dt;
  K->phys.inv_dt = 1.0f / c->sim_dt;
  K->phys.gdt = c->gravity * c->sim_dt;
  K->phys.half_dt = 0.5f * c->sim_dt;
  K->phys.im4 = 4.0f / c->body_mass;
  K->phys.imq = 0.25f * K->phys.im4;  // exact: the inverse mass seen by the average over the 4 contacts
  K->phys.J4x = 4.0f / c->body_inertia[0];
  K->phys.J4y = 4.0f / c->body_inertia[1];
  K->phys.J4z = 4.0f / c->body_inertia[2];
  K->phys.dt_over_Is = c->sim_dt / c->wheel_spin_inertia;
  K->phys.dt_over_Iy = c->sim_dt / c->body_inertia[1];
  K->phys.rw = c->wheel_radius;
  K->phys.rw_over_Is = c->wheel_radius / c->wheel_spin_inertia;
  K->phys.rw2_over_Is = (c->wheel_radius * c->wheel_radius) / c->wheel_spin_inertia;
  K->phys.beta_inv_dt = c->baumgarte * K->phys.inv_dt;
  K->phys.vdep = c->max_depenetration_velocity;
  K->phys.maxw = c->max_angular_velocity;
  K->phys.maxw2 = c->max_angular_velocity * c->max_angular_velocity;
  K->phys.damping = c->wheel_damping;
  K->phys.wlim = c->wheel_limit_velocity;
  K->phys.climit = c->wheel_limit_damping;
  K->phys.wx = c->wheel_pos[0];
  K->phys.wy = c->wheel_pos[1];
  K->phys.cz = c->wheel_pos[2] - c->com_height;
  K->task.com_height = c->com_height;
  K->phys.inv4wx = 1.0f / (4.0f * c->wheel_pos[0]);
  K->phys.inv4wy = 1.0f / (4.0f * c->wheel_pos[1]);
  K->phys.iters = c->solver_iterations;
  K->phys.nsub = c->decimation + c->extra_substeps;
  K->task.Kp = c->Kp; K->task.Kp_omega = c->Kp_omega; K->task.vehicle_mass = c->vehicle_mass; K->task.vehicle_inertia = c->vehicle_inertia;
  K->task.policy_dt = c->policy_dt; K->task.action_scale = c->action_scale; K->task.torque_clip = c->torque_clip;
  K->task.w1 = c->w1; K->task.w2 = c->w2; K->task.w3 = c->w3; K->task.sparse_bonus = c->sparse_bonus; K->task.sparse_threshold = c->sparse_threshold;
  K->task.terminal_reward = c->terminal_reward; K->task.max_sim_steps = c->max_sim_steps;
  K->task.gamma1_const = c->gamma1_const; K->task.gamma2_const = c->gamma2_const; K->task.bounds = c->bounds; K->task.clip_obs = c->clip_obs;
  K->reset.episode_length_s = c->episode_length_s; K->reset.eval_ramp_steps = c->eval_ramp_steps;
  K->reset.cmd_x_lo = c->cmd_x_lo; K->reset.cmd_x_hi = c->cmd_x_hi; K->reset.cmd_y_lo = c->cmd_y_lo; K->reset.cmd_y_hi = c->cmd_y_hi;
  K->reset.cmd_yaw_lo = c->cmd_yaw_lo; K->reset.cmd_yaw_hi = c->cmd_yaw_hi; K->reset.spawn_xy_range = c->spawn_xy_range;
  K->reset.cmd_x_span = c->cmd_x_span; K->reset.cmd_y_span = c->cmd_y_span; K->reset.cmd_yaw_span = c->cmd_yaw_span;
  K->reset.init_pos0 = c->init_pos[0]; K->reset.init_pos1 = c->init_pos[1]; K->reset.init_pos2 = c->init_pos[2];
  K->reset.init_quat0 = c->init_quat[0]; K->reset.init_quat1 = c->init_quat[1]; K->reset.init_quat2 = c->init_quat[2]; K->reset.init_quat3 = c->init_quat[3];
  K->reset.init_lin_vel0 = c->init_lin_vel[0]; K->reset.init_lin_vel1 = c->init_lin_vel[1]; K->reset.init_lin_vel2 = c->init_lin_vel[2];
  K->reset.init_ang_vel0 = c->init_ang_vel[0]; K->reset.init_ang_vel1 = c->init_ang_vel[1]; K->reset.init_ang_vel2 = c->init_ang_vel[2];
  K->reset.init_dof_vel0 = c->init_dof_vel[0]; K->reset.init_dof_vel1 = c->init_dof_vel[1];
  K->reset.init_dof_vel2 = c->init_dof_vel[2]; K->reset.init_dof_vel3 = c->init_dof_vel[3];
  K->task.action_dim = c->action_dim; K->task.mix_mode = c->mix_mode; K->task.gamma1_mode = c->gamma1_mode; K->task.gamma2_mode = c->gamma2_mode;
  K->task.guiding_mode = c->guiding_mode; K->task.cmd_sampler = c->cmd_sampler; K->task.eval_sums = c->eval_sums;
  K->task.max_episode_length = c->max_episode_length;
  K->task.env0_reward_guard = c->env0_reward_guard;
  K->task.seed_lo = (uint32_t)(c->seed & 0xffffffffu);
  K->task.seed_hi = (uint32_t)(c->seed >> 32);
  K->task.arena_offset = c->arena_offset;
  K->reset.cmd_dt = (double)c->decimation * (double)c->sim_dt;  // self.dt, a Python double (KAMMA.py:260)
  K->reset.cmd_episode_s = (double)c->episode_length_s;
  K->reset.cmd_thr_high = c->curriculum_threshold_high;
}

}  // namespace

struct ois_env {
  ois_config cfg;
  DevConsts K;
  DevBuffers B;
  Layout L;
  char* ws;
  int N, device, grid;
  int lpa;  // lanes per arena of the step kernel (1 | 2), chosen by pick_lpa()
  int lpa_forced;  // OIS_LANES_PER_ARENA: 0 = choose by size
  int simds;
  bool flat;    // every arena stands on the default plane n = (0, 0, 1): no ois_randomize_params slopes since ois_create /
                // the last ois_load_state of a flat blob (OIS_FLAT=0 forces the general kernels)
  bool flat_allowed;
  bool packed;  // LPA = 1 only: per-wheel math on v_pk_*_f32 (ois_physics_pk.h)
  bool waves2;  // plane kernel shell without the one-wave-per-SIMD claim (step_kernel_w2), chosen by arena count
  uint64_t seed;  // seed of the last ois_reset (RNG key of the episode stream)
  bool reset_done;
  double steps_done;
  // FIFO transition memory (ois_replay_*): caller-owned device memory, cursor on the host
  char* rp_mem;
  size_t rp_off[OIS_RP_COUNT_], rp_bytes[OIS_RP_COUNT_];
  int rp_head;
  bool rp_filled;
  ois_terrain tdesc;  // descriptor of the attached terrain (B.ter.on says whether one is)
  // ois_profile_*: start / stop events attached to the step kernel's own dispatch packets
  bool prof_on;
  std::vector<hipEvent_t> prof_ev;  // pairs; [2 i] start, [2 i + 1] stop
  size_t prof_used;                 // pairs handed to launches since ois_profile_begin
  double prof_ms;                   // kernel time of pairs already folded in (pool wrap-around)
  long long prof_launches;
  // ois_logs_*: the step's 32 + 32 log values in pinned host memory, published by logs_kernel
  float* logs_host;              // hipHostMalloc'ed (coherent): [32] episode info | [32] env-0 diagnostics | uint64 sequence
  unsigned long long logs_seq;   // sequence number of the last logs_kernel enqueued with the host block
  bool logs_on;                  // ois_step enqueues logs_kernel behind the step kernel
  bool logs_current;             // the last enqueued logs_kernel saw the current state
  float level_mean;              // torch.mean(terrain_levels) of a terrain whose levels cannot move (no curriculum), else < 0
  bool logs_force;               // state changed by something other than a step since the last logs kernel (reset, state blob,
                                 // terrain): the record on the device may be stale although its stamp is the newest
};

// Lanes per arena of the step kernel.
//  1: a lane owns an arena (two packed axle pairs). The layout for every arena count that gives each SIMD a
//     wave of its own: 24.4 us per policy step from 4 096 to 65 536 arenas (fused rollouts).
//  2: a lane PAIR owns an arena, each lane one packed side pair, the chassis replicated, four-wheel sums through
//     one DPP add (substep_pk2): 431 instead of 614 VALU instructions per lane and substep, 18.3 us per policy
//     step — while the doubled wave count still fits one wave per SIMD, i.e. up to simds * 32 arenas (32 768 on
//     an MI355X); at 65 536 arenas the pairs put two waves on every SIMD and the step takes 32.6 us.
// Both are bit-identical to the oracle. The terrain contact needs all four wheels in one lane.
static void pick_lpa(ois_env* e) {
  if (e->B.ter.on) e->lpa = 1;
  else if (e->lpa_forced) e->lpa = e->lpa_forced;
  else e->lpa = (e->packed && (long)e->N * 2 <= (long)e->simds * kBlock) ? 2 : 1;
}

namespace {
// scratch of ois_terrain_attach: the 16-bit min-field in 8 x 8 tiles | the 8-bit one in 16 x 8 tiles | its tile bases
size_t minfield_layout(int rows, int cols, size_t off[3]) {
  const size_t tpr = (size_t)((cols + 7) >> 3);
  const size_t t8 = (size_t)((rows + 7) >> 3) * tpr, t16 = (size_t)((rows + 15) >> 4) * tpr;
  off[0] = 0;
  off[1] = align256(t8 * 128);
  off[2] = off[1] + align256(t16 * 128);
  // the bases in blocks of 8 x 8 tiles (one line): the tiles under one robot's scan are 1-4 blocks
  const size_t blocks = (size_t)((((rows + 15) >> 4) + 7) >> 3) * ((tpr + 7) >> 3);
  return off[2] + align256(blocks * 64 * sizeof(int16_t));
}
size_t replay_layout(int N, int M, size_t off[OIS_RP_COUNT_], size_t bytes[OIS_RP_COUNT_]) {
  const size_t rows = (size_t)M * (size_t)N;
  const size_t sz[OIS_RP_COUNT_] = {rows * 6 * sizeof(float), rows * 4 * sizeof(float), rows * sizeof(float),
                                    rows * 6 * sizeof(float), rows};
  size_t o = 0;
  for (int i = 0; i < OIS_RP_COUNT_; ++i) { off[i] = o; bytes[i] = sz[i]; o = align256(o + sz[i]); }
  return o;
}
}  // namespace

// error sink shared with the other translation units of the library (ois_gae.hip); not exported
extern "C" __attribute__((visibility("hidden"))) int ois_fail_(int code, const char* msg) { return fail(code, "%s", msg); }

extern "C" {

int ois_abi_version(void) { return OIS_ABI_VERSION; }
const char* ois_last_error(void) { return g_err; }

int ois_config_default(ois_config* c) {
  if (!c) return fail(OIS_EINVAL, "null config");
  std::memset(c, 0, sizeof *c);
  c->struct_bytes = sizeof(ois_config);
  c->num_envs = 1;  // every reference script forces numEnvs = 1 (train_kamma_4d_nofifo_random_RLIL.py:79)
  c->arena_offset = 0;
  c->seed = 42;  // KAMMA.py:35
  c->action_dim = 4; c->mix_mode = OIS_MIX_SWITCH;
  c->gamma1_mode = c->gamma2_mode = OIS_GAMMA_SCHEDULE;
  c->gamma1_const = c->gamma2_const = 1.0f;
  c->guiding_mode = OIS_GUIDE_EXECUTED; c->cmd_sampler = OIS_CMD_RAND;
  c->max_sim_steps = 500000.0f;  // KAMMA_medium.py:179
  c->eval_sums = 0;
  c->sim_dt = 0.01f; c->decimation = 10; c->extra_substeps = 10;
  c->policy_dt = (float)(10 * 0.01);
  c->max_episode_length = (int)(10.0 / (10 * 0.01) + 0.5);
  c->episode_length_s = 10.0f;
  c->Kp = 0.035f;  // train_kamma_4d_nofifo_random_RLIL.py:63 (--stiffness)
  c->Kp_omega = 0.665f; c->vehicle_mass = 25.0f; c->vehicle_inertia = 1.05f;
  c->action_scale = 3.0f; c->torque_clip = 4.0f;
  c->w1 = -25.0f; c->w2 = -0.02f; c->w3 = -0.09f; c->sparse_bonus = 0.2f; c->sparse_threshold = 0.01f;
  c->terminal_reward = 0.0f;
  c->cmd_x_lo = 1.0f; c->cmd_x_hi = 2.0f;  // --linear_x [1,2]
  c->cmd_y_lo = -0.5f; c->cmd_y_hi = 0.5f; c->cmd_yaw_lo = 1.0f; c->cmd_yaw_hi = 1.1f;
  c->cmd_x_span = (float)(2.0 - 1.0); c->cmd_y_span = (float)(0.5 - -0.5); c->cmd_yaw_span = (float)(1.1 - 1.0);
  c->init_pos[0] = 0.0f; c->init_pos[1] = 0.0f; c->init_pos[2] = 0.1f;
  c->init_quat[0] = 1.0f;
  c->spawn_xy_range = 0.5f; c->bounds = 50.0f; c->clip_obs = 1000.0f; c->eval_ramp_steps = 50000.0f;
  // composite body of src/my_assets/urdf/origin_sym.urdf (see options_in_simulation_amd/config.py)
  const double mc = 21.0, mw = 1.0, wx = 0.205, wy = 0.25, wz = 0.056, iw = 0.01;
  const double m = mc + 4.0 * mw, com = 4.0 * mw * wz / m, dz = wz - com;
  c->gravity = 9.81f;
  c->body_mass = (float)m;
  c->body_inertia[0] = (float)(0.4 + mc * com * com + 4.0 * (mw * (wy * wy + dz * dz) + iw));
  c->body_inertia[1] = (float)(0.77 + mc * com * com + 4.0 * (mw * (wx * wx + dz * dz)));
  c->body_inertia[2] = (float)(1.01 + 4.0 * (mw * (wx * wx + wy * wy) + iw));
  c->com_height = (float)com;
  c->wheel_pos[0] = (float)wx; c->wheel_pos[1] = (float)wy; c->wheel_pos[2] = (float)wz;
  c->wheel_radius = 0.1175f; c->wheel_spin_inertia = 0.01f;
  // calibrated on the shipped training log, not parity (tools/calibrate_physics.py, tests/golden/ref_stats.json)
  c->wheel_damping = 0.0f; c->wheel_limit_velocity = 10.0f /* urdf:117 */; c->wheel_limit_damping = 0.6f;
  c->friction = 0.85f; c->baumgarte = 0.2f; c->max_depenetration_velocity = 100.0f;
  c->max_angular_velocity = 64.0f; c->solver_iterations = 4;
  c->curriculum_threshold_high = 10.0f;  // KAMMA.py:269
  return OIS_OK;
}

size_t ois_workspace_bytes(const ois_config* cfg) {
  if (validate(cfg) != OIS_OK) return 0;
  return make_layout(cfg->num_envs).total;
}

int ois_create(const ois_config* cfg, int device, void* workspace_device, size_t workspace_bytes, ois_env** out) {
  if (!out) return fail(OIS_EINVAL, "null out pointer");
  *out = nullptr;
  int rc = validate(cfg);
  if (rc != OIS_OK) return rc;
  if (!workspace_device) return fail(OIS_EINVAL, "null workspace");
  if (((uintptr_t)workspace_device & 255u) != 0) return fail(OIS_EINVAL, "workspace must be 256-byte aligned");
  Layout L = make_layout(cfg->num_envs);
  if (workspace_bytes < L.total) return fail(OIS_ENOMEM, "workspace smaller than ois_workspace_bytes()");
  int ndev = 0;
  HIP_TRY(hipGetDeviceCount(&ndev));
  if (device < 0 || device >= ndev) return fail(OIS_EINVAL, "device ordinal out of range");
  OIS_ON_DEVICE(device);
  ois_env* e = new (std::nothrow) ois_env;
  if (!e) return fail(OIS_ENOMEM, "host allocation failed");
  e->cfg = *cfg;
  derive(cfg, &e->K);
  e->L = L;
  e->ws = (char*)workspace_device;
  e->N = cfg->num_envs;
  e->device = device;
  e->grid = (e->N + kBlock - 1) / kBlock;
  {
    // Lanes per arena: see pick_lpa() below. OIS_LANES_PER_ARENA=1|2 overrides the choice.
    const char* w = getenv("OIS_LANES_PER_ARENA");
    e->lpa_forced = (w && (w[0] == '1' || w[0] == '2')) ? w[0] - '0' : 0;
    // Packed fp32 over wheel pairs is the default at every size: two thirds of the VALU instructions
    // per substep (614 vs 1 000), and the advantage holds with one wave per SIMD (29.9 vs 39.5 us
    // per policy step at 65 536 arenas, tools/terrain_probe.py) as with two (131 072 arenas and up).
    // OIS_PACKED=0 selects the scalar kernel.
    const char* pk = getenv("OIS_PACKED");
    e->packed = pk ? (pk[0] == '1') : true;
    const char* fl = getenv("OIS_FLAT");
    e->flat_allowed = fl ? (fl[0] == '1') : true;
    e->flat = e->flat_allowed;  // params start as (mu, 0, 0, 1) below
    // Plane kernel shell (see step_kernel / step_kernel_w2): two waves per SIMD allowed only where a
    // second wave per SIMD exists to be had, i.e. more arenas than one wave per SIMD covers.
    int simds = 1024;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) simds = 4 * prop.multiProcessorCount;
    e->simds = simds;
    const char* wv = getenv("OIS_STEP_WAVES");
    e->waves2 = wv ? (wv[0] == '2') : ((long)e->N > (long)simds * kBlock);
  }
  std::memset(&e->B.ter, 0, sizeof e->B.ter);
  pick_lpa(e);
  e->seed = cfg->seed;
  e->reset_done = false;
  e->steps_done = 0.0;
  char* w = e->ws;
  DevBuffers& B = e->B;
  B.pose = (float4*)(w + L.off[OIS_BUF_POSE]); B.quat = (float4*)(w + L.off[OIS_BUF_QUAT]);
  B.linvel = (float4*)(w + L.off[OIS_BUF_LINVEL]); B.angvel = (float4*)(w + L.off[OIS_BUF_ANGVEL]);
  B.wheel_vel = (float4*)(w + L.off[OIS_BUF_WHEEL_VEL]); B.commands = (float4*)(w + L.off[OIS_BUF_COMMANDS]);
  B.counters = (int4*)(w + L.off[OIS_BUF_COUNTERS]);
  B.sums_a = (float4*)(w + L.off[OIS_BUF_SUMS_A]); B.sums_b = (float4*)(w + L.off[OIS_BUF_SUMS_B]);
  B.sums_c = (float4*)(w + L.off[OIS_BUF_SUMS_C]); B.params = (float4*)(w + L.off[OIS_BUF_PARAMS]);
  B.lam_n = (uint2*)(w + L.off[OIS_BUF_LAMBDA_N]); B.lam_l = (uint2*)(w + L.off[OIS_BUF_LAMBDA_L]);
  B.lam_t = (uint2*)(w + L.off[OIS_BUF_LAMBDA_T]); B.wheel_torqs = (float4*)(w + L.off[OIS_BUF_WHEEL_TORQS]);
  B.obs = (float*)(w + L.off[OIS_BUF_OBS]); B.rew = (float*)(w + L.off[OIS_BUF_REW]);
  B.done = (uint8_t*)(w + L.off[OIS_BUF_DONE]); B.timeout = (uint8_t*)(w + L.off[OIS_BUF_TIMEOUT]);
  B.episode_info = (float*)(w + L.off[OIS_BUF_EPISODE_INFO]); B.env0_diag = (float*)(w + L.off[OIS_BUF_ENV0_DIAG]);
  B.partials = (float*)(w + L.partials); B.stamps = (int32_t*)(w + L.stamps); B.stat_acc = (double*)(w + L.stat_acc);
  B.rp_states = B.rp_actions = B.rp_rewards = B.rp_next = nullptr; B.rp_done = nullptr; B.rp_size = 0;
  B.env_origins = (float4*)(w + L.off[OIS_BUF_ENV_ORIGINS]); B.terrain_idx = (int4*)(w + L.off[OIS_BUF_TERRAIN_IDX]);
  B.cmd_fraction = (float*)(w + L.off[OIS_BUF_CMD_FRACTION]); B.reset_stamp = (int32_t*)(w + L.reset_stamp);
  std::memset(&B.ter, 0, sizeof B.ter);
  e->rp_mem = nullptr; e->rp_head = 0; e->rp_filled = false;
  e->prof_on = false; e->prof_used = 0; e->prof_ms = 0.0; e->prof_launches = 0;
  e->logs_host = nullptr; e->logs_seq = 0; e->logs_on = false; e->logs_current = false; e->logs_force = true; e->level_mean = -1.0f;
  // default per-arena parameters: mu = cfg.friction, flat plane n = +z
  hipError_t he = hipMemsetAsync(e->ws, 0, L.total, nullptr);
  if (he == hipSuccess) {
    hipLaunchKernelGGL(init_params_kernel, dim3((e->N + 255) / 256), dim3(256), 0, nullptr, e->B, e->N, cfg->friction);
    he = hipGetLastError();
  }
  if (he == hipSuccess) he = hipStreamSynchronize(nullptr);
  if (he != hipSuccess) {
    snprintf(g_err, sizeof g_err, "workspace initialisation failed: %s", hipGetErrorString(he));
    delete e;
    return OIS_EHIP;
  }
  *out = e;
  return OIS_OK;
}

void ois_destroy(ois_env* env) {
  if (!env) return;
  for (hipEvent_t ev : env->prof_ev) (void)hipEventDestroy(ev);
  if (env->logs_host) (void)hipHostFree(env->logs_host);
  delete env;
}

int ois_profile_begin(ois_env* e) {
  if (!e) return fail(OIS_EINVAL, "null env");
  e->prof_on = true; e->prof_used = 0; e->prof_ms = 0.0; e->prof_launches = 0;
  return OIS_OK;
}

int ois_profile_end(ois_env* e, double* kernel_ms_total, long long* launches) {
  if (!e || !kernel_ms_total || !launches) return fail(OIS_EINVAL, "null argument");
  if (!e->prof_on) return fail(OIS_ESTATE, "ois_profile_end without ois_profile_begin");
  OIS_ON_DEVICE(e->device);
  const size_t live = e->prof_used < e->prof_ev.size() / 2 ? e->prof_used : e->prof_ev.size() / 2;
  double ms_total = e->prof_ms;
  for (size_t i = 0; i < live; ++i) {
    float ms = 0.0f;
    HIP_TRY(hipEventSynchronize(e->prof_ev[2 * i + 1]));
    HIP_TRY(hipEventElapsedTime(&ms, e->prof_ev[2 * i], e->prof_ev[2 * i + 1]));
    ms_total += ms;
  }
  *kernel_ms_total = ms_total; *launches = e->prof_launches;
  e->prof_on = false; e->prof_used = 0; e->prof_ms = 0.0; e->prof_launches = 0;
  return OIS_OK;
}

int ois_reset(ois_env* e, uint64_t seed, void* stream) {
  if (!e) return fail(OIS_EINVAL, "null env");
  OIS_ON_DEVICE(e->device);
  e->seed = seed;
  e->K.task.seed_lo = (uint32_t)(seed & 0xffffffffu);
  e->K.task.seed_hi = (uint32_t)(seed >> 32);
  hipLaunchKernelGGL(reset_kernel, dim3(e->grid), dim3(kBlock), 0, (hipStream_t)stream, e->K, e->B, e->N);
  HIP_TRY(hipGetLastError());
  e->reset_done = true;
  e->steps_done = 0.0;
  e->logs_current = false; e->logs_force = true;
  return OIS_OK;
}

// next (start, stop) event pair of the profiling pool; a wrapped pool folds its oldest pair in first
static int prof_next_pair(ois_env* e, hipEvent_t* a, hipEvent_t* b) {
  constexpr size_t kPairs = 4096;
  const size_t i = e->prof_used % kPairs;
  if (e->prof_used >= kPairs) {  // reuse: the pair was recorded kPairs launches ago
    float ms = 0.0f;
    HIP_TRY(hipEventSynchronize(e->prof_ev[2 * i + 1]));
    HIP_TRY(hipEventElapsedTime(&ms, e->prof_ev[2 * i], e->prof_ev[2 * i + 1]));
    e->prof_ms += ms;
  } else if (e->prof_ev.size() < 2 * (i + 1)) {
    hipEvent_t x, y;
    HIP_TRY(hipEventCreate(&x));
    HIP_TRY(hipEventCreate(&y));
    e->prof_ev.push_back(x); e->prof_ev.push_back(y);
  }
  *a = e->prof_ev[2 * i]; *b = e->prof_ev[2 * i + 1];
  e->prof_used += 1; e->prof_launches += 1;
  return OIS_OK;
}

static int launch_kernel(ois_env* e, int k, int policy, const float* actions, float* reward_sum, void* stream) {
  OIS_ON_DEVICE(e->device);
  if (e->B.rp_size > 0 || e->prof_on) {
    // the transition-memory cursor and the profiling events live on the host and advance per CALL: a
    // captured launch would write the same memory row on every replay of the graph
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing((hipStream_t)stream, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone)
      return fail(OIS_ESTATE, e->prof_on ? "ois_profile_* is not for use under stream capture"
                                         : "a transition memory is attached: its cursor advances on the host per call, so a step that "
                                           "writes it cannot be captured into a HIP graph (detach it, or keep the step out of the graph)");
  }
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  if (e->prof_on) {
    int rc = prof_next_pair(e, &ev0, &ev1);
    if (rc != OIS_OK) return rc;
  }
  // with profiling on, the events ride on the kernel's dispatch packet: their elapsed time is the
  // kernel's own begin -> end, what rocprofv3 --kernel-trace reports (no launch gap, no marker packets)
#define OIS_LAUNCH(KERNEL_, LPA_)                                                                                      \
  do {                                                                                                                 \
    const dim3 grid_((LPA_ * e->N + kBlock - 1) / kBlock), block_(kBlock);                                             \
    if (e->prof_on)                                                                                                    \
      hipExtLaunchKernelGGL((KERNEL_), grid_, block_, 0, (hipStream_t)stream, ev0, ev1, 0, e->K, e->B, actions,         \
                            reward_sum, e->N, k, policy, e->rp_head);                                                  \
    else                                                                                                               \
      hipLaunchKernelGGL((KERNEL_), grid_, block_, 0, (hipStream_t)stream, e->K, e->B, actions, reward_sum, e->N, k,    \
                         policy, e->rp_head);                                                                          \
  } while (0)
  // instantiations: lanes per arena (2 = experiment), packed wheel pairs, height-field terrain
  // (ois_terrain_attach refuses LPA = 2), solver iterations unrolled (the default 4) or run-time
  const bool it4 = e->K.phys.iters == 4;
  const bool flat = e->flat && e->packed && it4 && !e->B.ter.on;  // the FLAT shells exist for the default configuration
  if ((policy & 0x200) != 0) {  // continuation launch of a level-curriculum rollout (can_fold() said these shells exist)
    if (e->B.ter.on) { if (e->tdesc.footprint_stride == 2) OIS_LAUNCH((step_kernel_terrain<true, 4, true, 2>), 1); else OIS_LAUNCH((step_kernel_terrain<true, 4, true>), 1); }
    else if (e->lpa == 2) { if (flat) OIS_LAUNCH((step_kernel<2, true, 4, true, true>), 2); else OIS_LAUNCH((step_kernel<2, true, 4, false, true>), 2); }
    else if (e->waves2) { if (flat) OIS_LAUNCH((step_kernel_w2<true, 4, true, true>), 1); else OIS_LAUNCH((step_kernel_w2<true, 4, false, true>), 1); }
    else { if (flat) OIS_LAUNCH((step_kernel<1, true, 4, true, true>), 1); else OIS_LAUNCH((step_kernel<1, true, 4, false, true>), 1); }
  } else if (flat) {
    if (e->lpa == 2) OIS_LAUNCH((step_kernel<2, true, 4, true>), 2);
    else if (e->waves2) OIS_LAUNCH((step_kernel_w2<true, 4, true>), 1);
    else OIS_LAUNCH((step_kernel<1, true, 4, true>), 1);
  } else if (e->lpa == 2) {
    if (e->packed) { if (it4) OIS_LAUNCH((step_kernel<2, true, 4>), 2); else OIS_LAUNCH((step_kernel<2, true, 0>), 2); }
    else OIS_LAUNCH((step_kernel<2, false, 0>), 2);
  }
  else if (e->B.ter.on && e->tdesc.footprint_stride == 2) {
    if (e->packed) { if (it4) OIS_LAUNCH((step_kernel_terrain<true, 4, false, 2>), 1); else OIS_LAUNCH((step_kernel_terrain<true, 0, false, 2>), 1); }
    else           { if (it4) OIS_LAUNCH((step_kernel_terrain<false, 4, false, 2>), 1); else OIS_LAUNCH((step_kernel_terrain<false, 0, false, 2>), 1); }
  }
  else if (e->B.ter.on) {
    if (e->packed) { if (it4) OIS_LAUNCH((step_kernel_terrain<true, 4>), 1); else OIS_LAUNCH((step_kernel_terrain<true, 0>), 1); }
    else           { if (it4) OIS_LAUNCH((step_kernel_terrain<false, 4>), 1); else OIS_LAUNCH((step_kernel_terrain<false, 0>), 1); }
  } else if (e->waves2) {
    if (e->packed) { if (it4) OIS_LAUNCH((step_kernel_w2<true, 4>), 1); else OIS_LAUNCH((step_kernel_w2<true, 0>), 1); }
    else           { if (it4) OIS_LAUNCH((step_kernel_w2<false, 4>), 1); else OIS_LAUNCH((step_kernel_w2<false, 0>), 1); }
  } else {
    if (e->packed) { if (it4) OIS_LAUNCH((step_kernel<1, true, 4>), 1); else OIS_LAUNCH((step_kernel<1, true, 0>), 1); }
    else           { if (it4) OIS_LAUNCH((step_kernel<1, false, 4>), 1); else OIS_LAUNCH((step_kernel<1, false, 0>), 1); }
  }
#undef OIS_LAUNCH
  HIP_TRY(hipGetLastError());
  e->steps_done += (double)e->N * (double)k;
  e->logs_current = false;
  if (e->B.rp_size > 0) {  // the kernel wrote rows rp_head .. rp_head + k - 1 (mod size)
    if (e->rp_head + k >= e->B.rp_size) e->rp_filled = true;
    e->rp_head = (e->rp_head + k) % e->B.rp_size;
  }
  return OIS_OK;
}

static int launch_steps(ois_env* e, int k, int policy, const float* actions, float* reward_sum, void* stream) {
  const bool level = e->K.task.cmd_sampler == OIS_CMD_LEVEL;
  if (!level && !(e->K.task.eval_sums && k > 1)) return launch_kernel(e, k, policy, actions, reward_sum, stream);
  // per-level command curriculum: the commands of all arenas are re-drawn between two policy steps
  // from a quantity that depends on every arena of the step before (level_command_kernel), so a
  // rollout is k launches of one step. The eval variants' Smoothness
  // sum has the same kind of dependence (old_torques of all arenas zeroed by any reset, KAMMA_eval.py:470):
  // k launches of one step
  // The command launch is only needed for what the caller can see: inside a rollout the NEXT step's launch does that work
  // in its prologue (bit 9, the FOLD shells of step_body), so a rollout of k steps is k + 1 launches instead of 2 k. The FOLD
  // shells exist for the packed, 4-iteration configuration (the default); any other keeps the launch pairs.
  const bool fold = level && e->packed && e->K.phys.iters == 4;
  for (int s = 0; s < k; ++s) {
    const float* a = (policy == OIS_POLICY_EXTERNAL) ? actions + (size_t)s * e->N * e->K.task.action_dim : actions;
    int rc = launch_kernel(e, 1, policy | (s > 0 ? 0x100 : 0) | ((fold && s > 0) ? 0x200 : 0), a, reward_sum, stream);
    if (rc != OIS_OK) return rc;
    if (level && (!fold || s == k - 1)) {
      hipLaunchKernelGGL(level_command_kernel, dim3((e->N + 255) / 256), dim3(256), 0, (hipStream_t)stream, e->K, e->B, e->N);
      HIP_TRY(hipGetLastError());
    }
  }
  return OIS_OK;
}

// extras["episode"] + the env-0 diagnostics of the current state: one launch (logs_kernel); with a host block attached
// the kernel also publishes them there under the next sequence number
static int enqueue_logs(ois_env* e, void* stream) {
  LogsArgs A;
  A.part = (float*)(e->ws + e->L.red_part);
  A.cmax = (int*)(e->ws + e->L.red_cmax);
  A.lvl = (long long*)(e->ws + e->L.red_lvl);
  A.level_mean = e->B.ter.on ? e->level_mean : -1.0f;
  A.host_block = e->logs_host;
  A.seq = e->logs_host ? e->logs_seq + 1 : 0;
  if (e->L.nchunks <= kBlockChunks && (!e->B.ter.on || e->level_mean >= 0.0f))  // one workgroup, no cross-XCD hop (logs_block_kernel)
    hipLaunchKernelGGL(logs_block_kernel, dim3(1), dim3(kLogsThreads), 0, (hipStream_t)stream, e->K, e->B, e->L.nwaves, e->L.nchunks, e->N, A,
                       e->logs_force ? 1 : 0);
  else {
    hipLaunchKernelGGL(logs_level1_kernel, dim3(e->L.nchunks), dim3(64), 0, (hipStream_t)stream, e->B, e->L.nwaves, e->N, A);
    hipLaunchKernelGGL(logs_level2_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, e->K, e->B, e->L.nchunks, e->N, A);
  }
  HIP_TRY(hipGetLastError());
  e->logs_force = false;
  if (e->logs_host) { e->logs_seq += 1; e->logs_current = true; }
  return OIS_OK;
}
static int logs_alloc(ois_env* e) {
  void* p = nullptr;
  // coherent (fine-grained) pinned memory: a store of the kernel is visible to the host when it lands, not at the kernel's end
  HIP_TRY(hipHostMalloc(&p, 4096, hipHostMallocCoherent | hipHostMallocMapped | hipHostMallocPortable));
  std::memset(p, 0, 4096);
  e->logs_host = (float*)p;
  e->logs_seq = 0;
  e->logs_current = false;
  return OIS_OK;
}

int ois_step(ois_env* e, const float* actions_device, void* stream) {
  if (!e) return fail(OIS_EINVAL, "null env");
  if (!actions_device) return fail(OIS_EINVAL, "null actions");
  if (!e->reset_done) return fail(OIS_ESTATE, "ois_step called before ois_reset");
  if (((uintptr_t)actions_device & (uintptr_t)(4 * e->K.task.action_dim - 1)) != 0)
    return fail(OIS_EINVAL, "actions must be aligned to action_dim * 4 bytes (one vector load per arena)");
  int rc = launch_steps(e, 1, OIS_POLICY_EXTERNAL, actions_device, nullptr, stream);
  if (rc != OIS_OK || !e->logs_on) return rc;
  // a caller that reads the step's logs after every step (ois_logs_enable): finalise and publish them right behind the
  // step kernel, so that the read finds them in host memory instead of enqueueing work into a drained stream
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing((hipStream_t)stream, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone) return OIS_OK;  // the sequence number is per call, not per replay
  OIS_ON_DEVICE(e->device);
  return enqueue_logs(e, stream);
}

int ois_rollout(ois_env* e, int k, int policy, const float* actions_device, float* reward_sum_device, void* stream) {
  if (!e) return fail(OIS_EINVAL, "null env");
  if (k < 1 || k > (1 << 20)) return fail(OIS_EINVAL, "k out of range");
  if (policy < OIS_POLICY_ZERO || policy > OIS_POLICY_EXTERNAL) return fail(OIS_EINVAL, "unknown policy");
  if (policy == OIS_POLICY_EXTERNAL && !actions_device) return fail(OIS_EINVAL, "EXTERNAL policy needs actions");
  if (policy == OIS_POLICY_EXTERNAL && ((uintptr_t)actions_device & (uintptr_t)(4 * e->K.task.action_dim - 1)) != 0)
    return fail(OIS_EINVAL, "actions must be aligned to action_dim * 4 bytes (one vector load per arena)");
  if (!e->reset_done) return fail(OIS_ESTATE, "ois_rollout called before ois_reset");
  return launch_steps(e, k, policy, actions_device, reward_sum_device, stream);
}

int ois_buffer(ois_env* e, int id, void** ptr_device, size_t* bytes) {
  if (!e || !ptr_device || !bytes) return fail(OIS_EINVAL, "null argument");
  if (id < 0 || id >= OIS_BUF_COUNT_) return fail(OIS_EINVAL, "unknown buffer id");
  *ptr_device = e->ws + e->L.off[id];
  *bytes = e->L.bytes[id];
  return OIS_OK;
}

int ois_randomize_params(ois_env* e, float mu_lo, float mu_hi, const float* slopes_host, int n_slopes, void* stream) {
  if (!e) return fail(OIS_EINVAL, "null env");
  if (n_slopes < 0 || n_slopes > 16 || (n_slopes > 0 && !slopes_host)) return fail(OIS_EINVAL, "bad slopes (0..16 values)");
  if (!(mu_lo >= 0.0f) || !(mu_hi >= mu_lo) || !(mu_hi < 1e6f)) return fail(OIS_EINVAL, "friction range: 0 <= mu_lo <= mu_hi (finite)");
  for (int k = 0; k < n_slopes; ++k)  // tan of the incline: the plane has to face up (n_z > 0) for a robot to be set down on it
    if (!(std::fabs(slopes_host[k]) <= 10.0f)) return fail(OIS_EINVAL, "slopes: |tan(incline)| <= 10, finite");
  OIS_ON_DEVICE(e->device);
  float* tbl = (float*)(e->ws + e->L.stat_out);  // scratch shared with ois_stats_fetch
  if (n_slopes > 0) HIP_TRY(hipMemcpyAsync(tbl, slopes_host, (size_t)n_slopes * sizeof(float), hipMemcpyHostToDevice, (hipStream_t)stream));
  DevConsts K = e->K;  // keyed by the CONFIG seed: arena parameters do not change with ois_reset(seed)
  K.task.seed_lo = (uint32_t)(e->cfg.seed & 0xffffffffu);
  K.task.seed_hi = (uint32_t)(e->cfg.seed >> 32);
  hipLaunchKernelGGL(randomize_params_kernel, dim3((e->N + 255) / 256), dim3(256), 0, (hipStream_t)stream, K, e->B, e->N,
                     mu_lo, mu_hi, tbl, n_slopes);
  HIP_TRY(hipGetLastError());
  if (n_slopes > 0) HIP_TRY(hipStreamSynchronize((hipStream_t)stream));  // slopes_host may be freed by the caller
  e->flat = e->flat_allowed && n_slopes == 0;  // without slopes the kernel wrote n = (0, 0, 1) for every arena
  return OIS_OK;
}

int ois_params_dirty(ois_env* e) {
  if (!e) return fail(OIS_EINVAL, "null env");
  e->flat = false;
  return OIS_OK;
}

int ois_kernel_info(const ois_env* e, char* out, size_t cap) {
  if (!e || !out || cap == 0) return fail(OIS_EINVAL, "null argument");
  const bool it4 = e->K.phys.iters == 4;
  const bool flat = e->flat && e->packed && it4 && !e->B.ter.on;
  const int nit = it4 ? 4 : 0, pk = e->packed ? 1 : 0;
  char id[64];
  if (e->B.ter.on) std::snprintf(id, sizeof id, "step_kernel_terrain<%d,%d>%s", pk, nit, e->tdesc.footprint_stride == 2 ? "+stride2" : "");
  else if (e->lpa == 2) std::snprintf(id, sizeof id, "step_kernel<2,%d,%d,%d>", pk, pk ? nit : 0, flat ? 1 : 0);
  else if (e->waves2) std::snprintf(id, sizeof id, "step_kernel_w2<%d,%d,%d>", pk, nit, flat ? 1 : 0);
  else std::snprintf(id, sizeof id, "step_kernel<1,%d,%d,%d>", pk, nit, flat ? 1 : 0);
  std::snprintf(out, cap, "%s lanes_per_arena=%d %s iterations=%s plane=%s%s", id, e->B.ter.on ? 1 : e->lpa,
                e->packed ? "packed" : "scalar", it4 ? "4(unrolled)" : "runtime", e->B.ter.on ? (e->tdesc.footprint_stride == 2 ? "height-field(footprints every 2nd substep)" : "height-field") : (flat ? "flat" : "per-arena"),
                e->K.task.cmd_sampler == OIS_CMD_LEVEL ? " +level_command_kernel" : "");
  return OIS_OK;
}

int ois_terrain_attach(ois_env* e, const ois_terrain* t, const int16_t* heights_device, int16_t* minfield_device,
                       const float* terrain_origins_device, const int32_t* levels_types_host, void* stream) {
  if (!e) return fail(OIS_EINVAL, "null env");
  OIS_ON_DEVICE(e->device);
  hipStream_t s = (hipStream_t)stream;
  const size_t f4 = (size_t)e->N * 4 * sizeof(float);
  e->logs_current = false; e->logs_force = true;  // the terrain-level mean of extras["episode"] changes with the attachment
  e->level_mean = -1.0f;
  if (!t) {  // detach: back to the per-arena plane, origins and indices cleared
    std::memset(&e->B.ter, 0, sizeof e->B.ter);
    e->K.task.cmd_sampler = e->cfg.cmd_sampler;
    pick_lpa(e);
    HIP_TRY(hipMemsetAsync(e->B.env_origins, 0, f4, s));
    HIP_TRY(hipMemsetAsync(e->B.terrain_idx, 0, f4, s));
    return OIS_OK;
  }
  if (t->struct_bytes != sizeof(ois_terrain)) return fail(OIS_EINVAL, "ois_terrain.struct_bytes does not match this library");
  if (t->rows < 2 || t->cols < 2 || !(t->horizontal_scale > 0.0f) || t->num_levels < 1 || t->num_types < 1 ||
      t->max_level < 0 || t->max_level >= t->num_levels || !heights_device || !minfield_device || !terrain_origins_device ||
      !levels_types_host)
    return fail(OIS_EINVAL, "bad terrain description");
  if (t->footprint_stride < 0 || t->footprint_stride > 2) return fail(OIS_EINVAL, "ois_terrain.footprint_stride must be 1 or 2 (0 = 1)");
  if (e->lpa_forced == 2) return fail(OIS_EINVAL, "the terrain contact needs one lane per arena (unset OIS_LANES_PER_ARENA)");
  // the *_eval files test evalsampling before curriculum (KAMMA_eval.py:544): their per-step re-draw would be the eval ramp
  if (t->curriculum && e->cfg.cmd_sampler == OIS_CMD_EVAL) return fail(OIS_EINVAL, "terrain.curriculum with the EvalSampling commands is not supported");
  size_t mf[3];
  if (minfield_layout(t->rows, t->cols, mf) >= ((size_t)1 << 31)) return fail(OIS_EINVAL, "height field too large (the kernels address it with 32-bit byte offsets)");
  if (((uintptr_t)minfield_device & 255u) != 0) return fail(OIS_EINVAL, "the min-field scratch must be 256-byte aligned");
  // per-arena origin = terrain_origins[level, type] (KAMMA.py:397): built on the host from the
  // caller's table, so the table is read back once
  const size_t no = (size_t)t->num_levels * t->num_types * 3;
  std::vector<float> tab(no), org((size_t)e->N * 4, 0.0f);
  std::vector<int32_t> idx((size_t)e->N * 4, 0);
  HIP_TRY(hipMemcpyAsync(tab.data(), terrain_origins_device, no * sizeof(float), hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  for (int i = 0; i < e->N; ++i) {
    const int lvl = levels_types_host[2 * i], typ = levels_types_host[2 * i + 1];
    if (lvl < 0 || lvl >= t->num_levels || typ < 0 || typ >= t->num_types) return fail(OIS_EINVAL, "terrain level / type out of range");
    idx[(size_t)4 * i] = lvl; idx[(size_t)4 * i + 1] = typ;
    for (int k = 0; k < 3; ++k) org[(size_t)4 * i + k] = tab[((size_t)lvl * t->num_types + typ) * 3 + k];
  }
  if (!t->curriculum) {  // the levels cannot move: their mean (KAMMA.py:488) is known here, as the same float the kernels would form
    long long lsum = 0;
    for (int i = 0; i < e->N; ++i) lsum += levels_types_host[2 * i];
    e->level_mean = (float)lsum / (float)e->N;
  }
  HIP_TRY(hipMemcpyAsync(e->B.env_origins, org.data(), f4, hipMemcpyHostToDevice, s));
  HIP_TRY(hipMemcpyAsync(e->B.terrain_idx, idx.data(), f4, hipMemcpyHostToDevice, s));
  HIP_TRY(hipStreamSynchronize(s));
  {
    const size_t cells = (size_t)t->rows * (size_t)t->cols;
    e->B.ter.tiles_per_row = (t->cols + 7) >> 3;
    hipLaunchKernelGGL(minfield_kernel, dim3((unsigned)((cells + 255) / 256)), dim3(256), 0, s, heights_device, minfield_device,
                       t->rows, t->cols);
    const unsigned ntiles = (unsigned)(((t->rows + 15) >> 4) * ((t->cols + 7) >> 3));
    hipLaunchKernelGGL(minfield8_kernel, dim3((ntiles + 63) / 64), dim3(64), 0, s, heights_device,
                       reinterpret_cast<uint8_t*>(minfield_device) + mf[1], reinterpret_cast<int16_t*>(reinterpret_cast<char*>(minfield_device) + mf[2]),
                       t->rows, t->cols);
    HIP_TRY(hipGetLastError());
  }
  DevTerrain& T = e->B.ter;
  T.h = heights_device; T.m = minfield_device; T.origins = terrain_origins_device;
  T.m8 = reinterpret_cast<const uint8_t*>(minfield_device) + mf[1];
  T.mb = reinterpret_cast<const int16_t*>(reinterpret_cast<const char*>(minfield_device) + mf[2]);
  T.rows = t->rows; T.cols = t->cols; T.ntypes = t->num_types; T.max_level = t->max_level;
  T.curriculum = t->curriculum; T.on = 1;
  T.hs = t->horizontal_scale; T.inv_hs = 1.0f / t->horizontal_scale; T.vs = t->vertical_scale; T.border = t->border_size;
  T.thr_lo = t->threshold_low; T.thr_hi = t->threshold_high;
  e->tdesc = *t;
  if (e->tdesc.footprint_stride != 2) e->tdesc.footprint_stride = 1;
  // terrain.curriculum also switches the commands to the per-level generators (KAMMA.py:533, 719)
  e->K.task.cmd_sampler = t->curriculum ? OIS_CMD_LEVEL : e->cfg.cmd_sampler;
  pick_lpa(e);
  if (t->curriculum && !(e->K.reset.cmd_thr_high > 0.0f)) e->K.reset.cmd_thr_high = 10.0f;
  return OIS_OK;
}

int ois_height_scan(ois_env* e, float* heights_out_device, void* stream) {
  if (!e || !heights_out_device) return fail(OIS_EINVAL, "null argument");
  if (!e->B.ter.on) return fail(OIS_ESTATE, "ois_height_scan needs an attached terrain");
  if (!e->reset_done) return fail(OIS_ESTATE, "ois_height_scan called before ois_reset");
  OIS_ON_DEVICE(e->device);
  hipLaunchKernelGGL(height_scan_kernel, dim3((unsigned)((e->N + kScanArenas - 1) / kScanArenas)), dim3(256), 0, (hipStream_t)stream,
                     e->K, e->B, e->N, heights_out_device);
  HIP_TRY(hipGetLastError());
  return OIS_OK;
}

int ois_push_robots(ois_env* e, void* stream) {
  if (!e) return fail(OIS_EINVAL, "null env");
  if (!e->reset_done) return fail(OIS_ESTATE, "ois_push_robots called before ois_reset");
  OIS_ON_DEVICE(e->device);
  hipLaunchKernelGGL(push_robots_kernel, dim3((e->N + 255) / 256), dim3(256), 0, (hipStream_t)stream, e->K, e->B, e->N);
  HIP_TRY(hipGetLastError());
  return OIS_OK;
}

int ois_outputs_copy(ois_env* e, float* obs_out, float* rew_out, uint8_t* terminated_out, uint8_t* truncated_out, void* stream) {
  if (!e) return fail(OIS_EINVAL, "null env");
  if (!e->reset_done) return fail(OIS_ESTATE, "ois_outputs_copy called before ois_reset");
  if (((uintptr_t)obs_out & 7u) != 0 || ((uintptr_t)rew_out & 3u) != 0) return fail(OIS_EINVAL, "obs_out must be 8-byte, rew_out 4-byte aligned");
  OIS_ON_DEVICE(e->device);
  hipLaunchKernelGGL(outputs_kernel, dim3((e->N + 255) / 256), dim3(256), 0, (hipStream_t)stream, e->B, e->N, obs_out, rew_out,
                     terminated_out, truncated_out);
  HIP_TRY(hipGetLastError());
  return OIS_OK;
}

int ois_episode_info(ois_env* e, void* stream) {
  if (!e) return fail(OIS_EINVAL, "null env");
  if (!e->reset_done) return fail(OIS_ESTATE, "ois_episode_info called before ois_reset");
  OIS_ON_DEVICE(e->device);
  return enqueue_logs(e, stream);
}

int ois_logs_enable(ois_env* e, int on) {
  if (!e) return fail(OIS_EINVAL, "null env");
  if (on && !e->logs_host) {
    OIS_ON_DEVICE(e->device);
    int rc = logs_alloc(e);
    if (rc != OIS_OK) return rc;
  }
  e->logs_on = on != 0;
  return OIS_OK;
}

int ois_logs_fetch(ois_env* e, float* episode_info_out, float* env0_diag_out, void* stream) {
  if (!e || !episode_info_out || !env0_diag_out) return fail(OIS_EINVAL, "null argument");
  if (!e->reset_done) return fail(OIS_ESTATE, "ois_logs_fetch called before ois_reset");
  OIS_ON_DEVICE(e->device);
  if (!e->logs_host) {
    int rc = logs_alloc(e);
    if (rc != OIS_OK) return rc;
  }
  if (!e->logs_current) {
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing((hipStream_t)stream, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone)
      return fail(OIS_ESTATE, "ois_logs_fetch waits on the host: not for use under stream capture");
    int rc = enqueue_logs(e, stream);
    if (rc != OIS_OK) return rc;
  }
  // the kernel stores the 64 values, then (system-scope release) the sequence number, into coherent pinned memory:
  // the host polls its own memory. hipStreamQuery is only the way out if the kernel never gets there.
  const unsigned long long want = e->logs_seq;
  const unsigned long long* flag = reinterpret_cast<const unsigned long long*>(e->logs_host + 2 * OIS_EPISODE_INFO_LEN);
  for (unsigned spins = 1;; ++spins) {
    if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == want) break;
    if ((spins & 0x3fffu) == 0u) {
      const hipError_t q = hipStreamQuery((hipStream_t)stream);
      if (q == hipSuccess) {
        if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == want) break;
        return fail(OIS_EHIP, "the stream drained without the logs kernel publishing (was ois_step given another stream?)");
      }
      if (q != hipErrorNotReady) HIP_TRY(q);
    }
    __builtin_ia32_pause();
  }
  std::memcpy(episode_info_out, e->logs_host, OIS_EPISODE_INFO_LEN * sizeof(float));
  std::memcpy(env0_diag_out, e->logs_host + OIS_EPISODE_INFO_LEN, OIS_ENV0_DIAG_LEN * sizeof(float));
  return OIS_OK;
}

int ois_stats_fetch(ois_env* e, ois_stats* out, int reset_after, void* stream) {
  if (!e || !out) return fail(OIS_EINVAL, "null argument");
  if (!e->reset_done) return fail(OIS_ESTATE, "ois_stats_fetch called before ois_reset");
  OIS_ON_DEVICE(e->device);
  double* d = (double*)(e->ws + e->L.stat_out);
  double* part8 = (double*)(e->ws + e->L.red_part8);
  hipLaunchKernelGGL(stats_level1_kernel, dim3(e->L.nchunks), dim3(64), 0, (hipStream_t)stream, e->B, e->L.nwaves, part8, reset_after);
  hipLaunchKernelGGL(stats_level2_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, e->L.nchunks, part8, d);
  HIP_TRY(hipGetLastError());
  double h[8];
  HIP_TRY(hipMemcpyAsync(h, d, sizeof h, hipMemcpyDeviceToHost, (hipStream_t)stream));
  HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
  out->episodes = h[0]; out->return_sqsum = h[1]; out->length_sum = h[2]; out->return_sum = h[3];
  out->return_min = -h[6]; out->return_max = h[7];
  out->steps = e->steps_done; out->reserved = 0.0;
  if (reset_after) e->steps_done = 0.0;
  return OIS_OK;
}

int ois_env0_diag(ois_env* e, float* host_out, void* stream) {
  if (!e || !host_out) return fail(OIS_EINVAL, "null argument");
  OIS_ON_DEVICE(e->device);
  HIP_TRY(hipMemcpyAsync(host_out, e->B.env0_diag, OIS_ENV0_DIAG_LEN * sizeof(float), hipMemcpyDeviceToHost, (hipStream_t)stream));
  HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
  return OIS_OK;
}

size_t ois_replay_bytes(const ois_env* e, int memory_size) {
  if (!e || memory_size < 1) return 0;
  size_t off[OIS_RP_COUNT_], bytes[OIS_RP_COUNT_];
  return replay_layout(e->N, memory_size, off, bytes);
}

int ois_replay_attach(ois_env* e, int memory_size, void* memory_device, size_t bytes) {
  if (!e) return fail(OIS_EINVAL, "null env");
  DevBuffers& B = e->B;
  if (!memory_device) {  // detach
    B.rp_states = B.rp_actions = B.rp_rewards = B.rp_next = nullptr; B.rp_done = nullptr; B.rp_size = 0;
    e->rp_mem = nullptr; e->rp_head = 0; e->rp_filled = false;
    return OIS_OK;
  }
  if (memory_size < 1 || memory_size > (1 << 24)) return fail(OIS_EINVAL, "memory_size out of range");
  if (((uintptr_t)memory_device & 255u) != 0) return fail(OIS_EINVAL, "replay memory must be 256-byte aligned");
  const size_t need = replay_layout(e->N, memory_size, e->rp_off, e->rp_bytes);
  if (bytes < need) return fail(OIS_ENOMEM, "replay memory smaller than ois_replay_bytes()");
  char* m = (char*)memory_device;
  e->rp_mem = m;
  B.rp_states = (float*)(m + e->rp_off[OIS_RP_STATES]); B.rp_actions = (float*)(m + e->rp_off[OIS_RP_ACTIONS]);
  B.rp_rewards = (float*)(m + e->rp_off[OIS_RP_REWARDS]); B.rp_next = (float*)(m + e->rp_off[OIS_RP_NEXT_STATES]);
  B.rp_done = (uint8_t*)(m + e->rp_off[OIS_RP_TERMINATED]);
  B.rp_size = memory_size;
  e->rp_head = 0; e->rp_filled = false;
  return OIS_OK;
}

int ois_replay_buffer(ois_env* e, int id, void** ptr_device, size_t* bytes) {
  if (!e || !ptr_device || !bytes) return fail(OIS_EINVAL, "null argument");
  if (!e->rp_mem) return fail(OIS_ESTATE, "no replay memory attached");
  if (id < 0 || id >= OIS_RP_COUNT_) return fail(OIS_EINVAL, "unknown replay buffer id");
  *ptr_device = e->rp_mem + e->rp_off[id];
  *bytes = e->rp_bytes[id];
  return OIS_OK;
}

int ois_replay_cursor(const ois_env* e, int* memory_index, int* filled) {
  if (!e || !memory_index || !filled) return fail(OIS_EINVAL, "null argument");
  *memory_index = e->rp_head;
  *filled = e->rp_filled ? 1 : 0;
  return OIS_OK;
}

struct StateHeader {
  uint32_t magic, abi;
  int32_t num_envs;
  int32_t reset_done;
  uint64_t seed;
  double steps_done;
  uint64_t payload;
  uint64_t config_hash;   // every field of ois_config: variant switches, arena_offset, task and physics constants
  uint64_t terrain_hash;  // descriptor of the attached terrain, 0 = none
  int32_t rp_size, rp_head, rp_filled;  // transition memory cursor (the memory itself is the caller's)
  int32_t cmd_sampler;    // effective sampler (a terrain with curriculum switches it)
  int32_t plane_flat;     // every params row holds n = (0, 0, 1) (ois_env::flat)
};

static uint64_t fnv1a(const void* p, size_t n, uint64_t h = 1469598103934665603ull) {
  const unsigned char* b = (const unsigned char*)p;
  for (size_t i = 0; i < n; ++i) { h ^= b[i]; h *= 1099511628211ull; }
  return h;
}
static uint64_t terrain_hash(const ois_env* e) {
  if (!e->B.ter.on) return 0;
  const ois_terrain& t = e->tdesc;
  const int32_t iv[7] = {t.rows, t.cols, t.num_levels, t.num_types, t.curriculum, t.max_level, t.footprint_stride};  // stride: 1 or 2 (attach)
  const float fv[5] = {t.horizontal_scale, t.vertical_scale, t.border_size, t.threshold_low, t.threshold_high};
  return fnv1a(fv, sizeof fv, fnv1a(iv, sizeof iv)) | 1ull;
}

int ois_replay_sample(ois_env* e, const int64_t* indexes_device, int count, float* states_out, float* actions_out,
                      float* rewards_out, float* next_states_out, uint8_t* terminated_out, void* stream) {
  if (!e || !indexes_device) return fail(OIS_EINVAL, "null argument");
  if (e->B.rp_size <= 0) return fail(OIS_ESTATE, "no replay memory attached");
  if (count < 1) return fail(OIS_EINVAL, "count must be >= 1");
  OIS_ON_DEVICE(e->device);
  const size_t rows = (size_t)e->B.rp_size * (size_t)e->N;
  hipLaunchKernelGGL(replay_sample_kernel, dim3((count + 255) / 256), dim3(256), 0, (hipStream_t)stream, e->B, rows, indexes_device,
                     count, states_out, actions_out, rewards_out, next_states_out, terminated_out);
  HIP_TRY(hipGetLastError());
  return OIS_OK;
}

size_t ois_minfield_bytes(int rows, int cols) {
  if (rows < 1 || cols < 1) return 0;
  size_t off[3];
  return minfield_layout(rows, cols, off);
}

size_t ois_state_bytes(const ois_env* e) { return e ? sizeof(StateHeader) + e->L.total : 0; }

int ois_save_state(ois_env* e, void* host_blob, size_t bytes, void* stream) {
  if (!e || !host_blob) return fail(OIS_EINVAL, "null argument");
  if (bytes < ois_state_bytes(e)) return fail(OIS_ENOMEM, "blob smaller than ois_state_bytes()");
  OIS_ON_DEVICE(e->device);
  StateHeader h;
  std::memset(&h, 0, sizeof h);
  h.magic = 0x3053494fu /* "OIS0" */; h.abi = OIS_ABI_VERSION; h.num_envs = e->N; h.reset_done = e->reset_done ? 1 : 0;
  h.seed = e->seed; h.steps_done = e->steps_done; h.payload = e->L.total;
  h.config_hash = fnv1a(&e->cfg, sizeof e->cfg); h.terrain_hash = terrain_hash(e);
  h.rp_size = e->B.rp_size; h.rp_head = e->rp_head; h.rp_filled = e->rp_filled ? 1 : 0; h.cmd_sampler = e->K.task.cmd_sampler;
  h.plane_flat = e->flat ? 1 : 0;
  std::memcpy(host_blob, &h, sizeof h);
  HIP_TRY(hipMemcpyAsync((char*)host_blob + sizeof h, e->ws, e->L.total, hipMemcpyDeviceToHost, (hipStream_t)stream));
  HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
  return OIS_OK;
}

int ois_load_state(ois_env* e, const void* host_blob, size_t bytes, void* stream) {
  if (!e || !host_blob) return fail(OIS_EINVAL, "null argument");
  if (bytes < sizeof(StateHeader)) return fail(OIS_EINVAL, "blob too small");
  StateHeader h;
  std::memcpy(&h, host_blob, sizeof h);
  if (h.magic != 0x3053494fu || h.abi != OIS_ABI_VERSION) return fail(OIS_EINVAL, "not an ois state blob of this ABI");
  if (h.num_envs != e->N || h.payload != e->L.total || bytes < sizeof h + h.payload) return fail(OIS_EINVAL, "state blob does not match this environment");
  if (h.config_hash != fnv1a(&e->cfg, sizeof e->cfg))
    return fail(OIS_EINVAL, "state blob was saved under a different ois_config (variant, arena_offset, task or physics constants)");
  if (h.terrain_hash != terrain_hash(e))
    return fail(OIS_EINVAL, h.terrain_hash ? "state blob was saved with a terrain attached: attach the same terrain (ois_terrain_attach) before ois_load_state"
                                           : "state blob was saved on the plane: detach the terrain before ois_load_state");
  if (h.rp_size != e->B.rp_size)
    return fail(OIS_EINVAL, h.rp_size ? "state blob was saved with a transition memory attached: attach one of the same size (its tensors are the caller's to restore) before ois_load_state"
                                      : "state blob was saved without a transition memory: detach it before ois_load_state");
  OIS_ON_DEVICE(e->device);
  HIP_TRY(hipMemcpyAsync(e->ws, (const char*)host_blob + sizeof h, e->L.total, hipMemcpyHostToDevice, (hipStream_t)stream));
  HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
  e->reset_done = h.reset_done != 0;
  e->logs_current = false; e->logs_force = true;
  e->level_mean = -1.0f;  // the terrain rows came with the blob: the level mean is summed on the device from now on
  e->seed = h.seed;
  e->K.task.seed_lo = (uint32_t)(h.seed & 0xffffffffu);
  e->K.task.seed_hi = (uint32_t)(h.seed >> 32);
  e->steps_done = h.steps_done;
  e->rp_head = h.rp_head; e->rp_filled = h.rp_filled != 0;
  e->flat = e->flat_allowed && h.plane_flat != 0;  // the params rows came with the blob
  return OIS_OK;
}

}  // extern "C"
