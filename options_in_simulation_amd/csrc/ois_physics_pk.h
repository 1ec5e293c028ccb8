// ois_physics_pk.h — the LPA = 1 substep with the per-wheel math on PACKED fp32.
//
// Same operations in the same order as substep<EXACT, 1> (ois_physics.h), so the results are
// bit-identical; only the instruction selection differs: the wheels of an axle — (fl, fr) and (rl, rr) — ride
// in the two halves of float2 operands, which hipcc lowers to v_pk_fma_f32 / v_pk_mul_f32 /
// v_pk_add_f32 (IEEE per element). Why it pays: at 65 536 arenas there is one wave per SIMD, and
// a lone wave issues one VALU instruction per ~4-5 cycles whether it is a v_fma_f32 or a
// v_pk_fma_f32 (tools/valu_rate.hip: 2.0-2.4 ns vs 2.5 ns), so a packed instruction does two
// wheels for the price of one: 614 instead of 1 000 VALU instructions per substep. With two
// waves per SIMD (more arenas) the gap narrows but stays.
// (In the earlier 4-lanes-per-arena layout, with 4 waves per SIMD, packed ran at half rate and
// bought nothing: DESIGN.md section 4.)
#pragma once
#include "ois_physics.h"

namespace ois {

typedef float f2 __attribute__((ext_vector_type(2)));

OIS_DEV f2 fma2(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
OIS_DEV f2 fma2(float a, f2 b, f2 c) { return __builtin_elementwise_fma((f2)(a), b, c); }
OIS_DEV f2 fma2(f2 a, float b, f2 c) { return __builtin_elementwise_fma(a, (f2)(b), c); }
OIS_DEV f2 fma2(f2 a, f2 b, float c) { return __builtin_elementwise_fma(a, b, (f2)(c)); }
OIS_DEV f2 fma2(f2 a, float b, float c) { return __builtin_elementwise_fma(a, (f2)(b), (f2)(c)); }
OIS_DEV f2 fma2(float a, f2 b, float c) { return __builtin_elementwise_fma((f2)(a), b, (f2)(c)); }
OIS_DEV f2 max2(f2 a, f2 b) { f2 r; r.x = fmaxf(a.x, b.x); r.y = fmaxf(a.y, b.y); return r; }
OIS_DEV f2 min2(f2 a, f2 b) { f2 r; r.x = fminf(a.x, b.x); r.y = fminf(a.y, b.y); return r; }
// pair j = wheels (j, j + 2): j = 0 front (fl, fr), j = 1 rear (rl, rr). With this pairing the
// SPEC's wheel sum (w0 + w1) + (w2 + w3) is one packed add {w0 + w1, w2 + w3} and one scalar add.
template <int N> OIS_DEV f2 ld2(const float (&a)[N], int j) { f2 r; r.x = a[j]; r.y = a[j + 2]; return r; }
template <int N> OIS_DEV void st2(float (&a)[N], int j, f2 v) { a[j] = v.x; a[j + 2] = v.y; }
OIS_DEV float sum4(const f2 (&c)[2]) { const f2 s = c[0] + c[1]; return s.x + s.y; }
// clamp(x, -lim, lim) with lim >= 0: one v_med3_f32 instead of v_max + v_min (same value for
// every non-NaN input)
OIS_DEV f2 clamp2(f2 x, f2 lim) {
  f2 r;
  r.x = __builtin_amdgcn_fmed3f(x.x, -lim.x, lim.x);
  r.y = __builtin_amdgcn_fmed3f(x.y, -lim.y, lim.y);
  return r;
}
OIS_DEV f2 exact_rcp2(f2 d) {  // exact_rcp (ois_physics.h) on both halves: 2 v_rcp_f32 + 6 v_pk_fma_f32
  f2 r0;
  r0.x = __builtin_amdgcn_rcpf(d.x); r0.y = __builtin_amdgcn_rcpf(d.y);
  const f2 e = fma2(-d, r0, 1.0f);
  const f2 r1 = fma2(e, r0, r0);
  const f2 g = fma2(-d, r1, 1.0f);
  const f2 q = fma2(g, r1, r1);
  const f2 h = fma2(-d, q, 1.0f);
  return fma2(h, r1, q);
}
OIS_DEV f2 newton_rcp2(f2 m, f2 k) {
  const f2 e = fma2(-k, m, 1.0f);
  return fma2(m, e, m);
}

// (The footprints and the plane fit of the height field stay on scalar instructions here — terrain_footprints / terrain_plane of
// ois_physics.h: their per-wheel arithmetic on the axle pairs was built in round 4, 26 instructions fewer per substep and
// bit-identical, and measured SLOWER on the same box: 47.8 vs 45.5 us per fused step, 58.0 vs 52.1 us per launch behind the
// actor's GEMMs. The four gathers then leave together, late, instead of one by one as each wheel's cell is known.)
// FLAT: the ground is the default plane n = (0, 0, 1) for every arena (no ois_randomize_params slopes, no terrain —
// the host knows): n_b is the third row of R and n.p is p_z; the 12 instructions that multiply by the zero components of n
// are not issued (SPEC 3.2: the same IEEE values, a zero may carry the other sign).
template <bool EXACT, bool TER = false, int NIT = 0, bool FLAT = false, int TMODE = kEverySubstep>
OIS_DEV void substep_pk(const PhysK& K, const DevTerrain& T, PhysT<1>& S) {
  static_assert(!(FLAT && TER), "the height field has its own plane");
  static_assert(TMODE == kEverySubstep || (TER && !EXACT), "footprint stride modes: later substeps on a height field");
#define OIS_PR _Pragma("unroll") for (int j = 0; j < 2; ++j)
  float R[9];
  rotmat(S.q, R);
  float vb[3], nb[3], wb[3] = {S.wb[0], S.wb[1], S.wb[2]};
  float n[3] = {S.n[0], S.n[1], S.n[2]}, off[4] = {0.0f, 0.0f, 0.0f, 0.0f};
  Footprints cur;
  if constexpr (TER) {  // one-substep-lagged footprints, see substep<> in ois_physics.h
    if constexpr (TMODE != kOddSubstep) terrain_footprints<EXACT>(K, T, S, R, cur);
    if (EXACT) footprints_store(S, cur);
    if constexpr (TMODE == kEvenSubstep) {  // the plane of the substep before (ois_physics.h substep<>)
      n[0] = S.pn[0]; n[1] = S.pn[1]; n[2] = S.pn[2];
      off[0] = S.poff[0]; off[1] = S.poff[1]; off[2] = S.poff[2]; off[3] = S.poff[3];
    } else {
      terrain_plane(K, S, n, off);
      if constexpr (TMODE == kOddSubstep) {
        S.pn[0] = n[0]; S.pn[1] = n[1]; S.pn[2] = n[2];
        S.poff[0] = off[0]; S.poff[1] = off[1]; S.poff[2] = off[2]; S.poff[3] = off[3];
      }
    }
  }
  rotT(R, S.v, vb);
  if constexpr (FLAT) { nb[0] = R[6]; nb[1] = R[7]; nb[2] = R[8]; }
  else rotT(R, n, nb);
  vb[0] = __builtin_fmaf(-K.gdt, R[6], vb[0]);
  vb[1] = __builtin_fmaf(-K.gdt, R[7], vb[1]);
  vb[2] = __builtin_fmaf(-K.gdt, R[8], vb[2]);
  f2 spin[2], tau[2], ln[2], l1[2], l2[2], mn[2], m1[2], m2[2];
  OIS_PR {
    spin[j] = ld2(S.spin, j);
    ln[j] = ld2(S.ln, j); l1[j] = ld2(S.l1, j); l2[j] = ld2(S.l2, j);
    mn[j] = ld2(S.mn, j); m1[j] = ld2(S.m1, j); m2[j] = ld2(S.m2, j);
  }
  tau[0].x = S.tau[0]; tau[0].y = S.tau[2]; tau[1].x = S.tau[1]; tau[1].y = S.tau[3];  // constant indices: keeps tau out of LDS
  // wheel drive + viscous joint damping, reaction on the chassis
  f2 Tq[2];
  OIS_PR {
    const f2 jv = spin[j] - wb[1];
    const f2 over = jv - clamp2(jv, (f2)(K.wlim));  // signed excess over the joint velocity limit
    Tq[j] = fma2(-K.climit, over, fma2(-K.damping, jv, tau[j]));
    spin[j] = fma2(K.dt_over_Is, Tq[j], spin[j]);
  }
  const float Tsum = sum4(Tq);
  wb[1] = __builtin_fmaf(-K.dt_over_Iy, Tsum, wb[1]);
  // contact frame shared by the four wheels (common axle direction = body y)
  float s2 = __builtin_fmaf(nb[2], nb[2], nb[0] * nb[0]);
  s2 = fmaxf(s2, 1e-12f);
  constexpr bool REX = EXACT || TER;  // every substep on a height field (ois_physics.h substep<>)
  const float inv_s = REX ? exact_rcp(sqrtf(s2)) : newton_rsqrt(S.inv_s, s2);  // s2 >= 1e-12
  S.inv_s = inv_s;
  const float dx = -(nb[0] * inv_s), dz = -(nb[2] * inv_s);
  const float t1x = -dz, t1z = dx;
  const float t2x = nb[1] * t1z;
  const float t2y = __builtin_fmaf(nb[2], t1x, -(nb[0] * t1z));
  const float t2z = -(nb[1] * t1x);
  const float np = FLAT ? S.p[2] : __builtin_fmaf(n[2], S.p[2], __builtin_fmaf(n[1], S.p[1], n[0] * S.p[0]));
  const float cz = K.cz;
  const f2 cy = {K.wy, -K.wy};  // (left, right) wheel of each axle (urdf:112-142: left = +y)
  const f2 ry = cy;
  const float rz = __builtin_fmaf(K.rw, dz, cz);
  f2 b[2], an[2][3], a1[2][3], a2[2][3], Jan[2][3], Ja1[2][3], Ja2[2][3];
  OIS_PR {
    const float cx = j ? -K.wx : K.wx;  // front axle +x, rear axle -x
    const float rx = __builtin_fmaf(K.rw, dx, cx);
    // gap = np + fma(nb2, rz, fma(nb1, ry, nb0 * rx))
    const f2 gap = (TER ? np - ld2(off, j) : (f2)(np)) + fma2(nb[2], (f2)(rz), fma2(nb[1], ry, nb[0] * rx));
    const f2 bA = -(gap * K.inv_dt), bB = -(gap * K.beta_inv_dt);  // min3 = the case split of substep<> (0 < beta < 1)
    f2 bb;
    bb.x = __builtin_fminf(__builtin_fminf(bA.x, bB.x), K.vdep);
    bb.y = __builtin_fminf(__builtin_fminf(bA.y, bB.y), K.vdep);
    b[j] = bb;
    an[j][0] = fma2(ry, nb[2], -(rz * nb[1]));
    an[j][1] = (f2)(__builtin_fmaf(rz, nb[0], -(rx * nb[2])));
    an[j][2] = fma2(rx, (f2)(nb[1]), -(ry * nb[0]));
    a1[j][0] = cy * t1z;
    a1[j][1] = (f2)(__builtin_fmaf(cz, t1x, -(cx * t1z)));
    a1[j][2] = -(cy * t1x);
    a2[j][0] = fma2(ry, t2z, -(rz * t2y));
    a2[j][1] = (f2)(__builtin_fmaf(rz, t2x, -(rx * t2z)));
    a2[j][2] = fma2(rx, (f2)(t2y), -(ry * t2x));
    Jan[j][0] = K.J4x * an[j][0]; Jan[j][1] = K.J4y * an[j][1]; Jan[j][2] = K.J4z * an[j][2];
    Ja1[j][0] = K.J4x * a1[j][0]; Ja1[j][1] = K.J4y * a1[j][1]; Ja1[j][2] = K.J4z * a1[j][2];
    Ja2[j][0] = K.J4x * a2[j][0]; Ja2[j][1] = K.J4y * a2[j][1]; Ja2[j][2] = K.J4z * a2[j][2];
    const f2 kn = fma2(an[j][2], Jan[j][2], fma2(an[j][1], Jan[j][1], fma2(an[j][0], Jan[j][0], K.im4)));
    const f2 k1 = fma2(a1[j][2], Ja1[j][2], fma2(a1[j][1], Ja1[j][1], fma2(a1[j][0], Ja1[j][0], K.im4))) + K.rw2_over_Is;
    const f2 k2 = fma2(a2[j][2], Ja2[j][2], fma2(a2[j][1], Ja2[j][1], fma2(a2[j][0], Ja2[j][0], K.im4)));
    if (REX) {
      mn[j] = exact_rcp2(kn); m1[j] = exact_rcp2(k1); m2[j] = exact_rcp2(k2);
    } else {
      mn[j] = newton_rcp2(mn[j], kn);
      m1[j] = newton_rcp2(m1[j], k1);
      m2[j] = newton_rcp2(m2[j], k2);
    }
  }
  // the solve carries the projections (un, u1, u2) of v_b on the contact frame (see substep<> in ois_physics.h)
  float un = __builtin_fmaf(nb[2], vb[2], __builtin_fmaf(nb[1], vb[1], nb[0] * vb[0]));
  float u1 = __builtin_fmaf(t1z, vb[2], t1x * vb[0]);
  float u2 = __builtin_fmaf(t2z, vb[2], __builtin_fmaf(t2y, vb[1], t2x * vb[0]));
  f2 ox[2], oy[2], oz[2];
  // warm start: one mass-split averaging pass with the cached impulses
  OIS_PR ox[j] = fma2(l2[j], Ja2[j][0], fma2(l1[j], Ja1[j][0], fma2(ln[j], Jan[j][0], wb[0])));
  OIS_PR oy[j] = fma2(l2[j], Ja2[j][1], fma2(l1[j], Ja1[j][1], fma2(ln[j], Jan[j][1], wb[1])));
  OIS_PR oz[j] = fma2(l2[j], Ja2[j][2], fma2(l1[j], Ja1[j][2], fma2(ln[j], Jan[j][2], wb[2])));
  OIS_PR spin[j] = fma2(-K.rw_over_Is, l1[j], spin[j]);
  {
    const float Sn = sum4(ln), S1 = sum4(l1), S2 = sum4(l2);
    un = __builtin_fmaf(K.imq, Sn, un); u1 = __builtin_fmaf(K.imq, S1, u1); u2 = __builtin_fmaf(K.imq, S2, u2);
  }
  wb[0] = 0.25f * sum4(ox); wb[1] = 0.25f * sum4(oy); wb[2] = 0.25f * sum4(oz);
  const int iters = NIT > 0 ? NIT : K.iters;
#pragma unroll
  for (int it = 0; it < iters; ++it) {
    f2 u[2], dl[2], nw[2], lim[2], dn[2], d1[2];
    // normal row
    OIS_PR u[j] = fma2(an[j][0], wb[0], un);
    OIS_PR u[j] = fma2(an[j][1], wb[1], u[j]);
    OIS_PR u[j] = fma2(an[j][2], wb[2], u[j]);
    OIS_PR nw[j] = max2(fma2(b[j] - u[j], mn[j], ln[j]), (f2)(0.0f));
    OIS_PR dn[j] = nw[j] - ln[j];
    OIS_PR ln[j] = nw[j];
    OIS_PR ox[j] = fma2(dn[j], Jan[j][0], wb[0]);
    OIS_PR oy[j] = fma2(dn[j], Jan[j][1], wb[1]);
    OIS_PR oz[j] = fma2(dn[j], Jan[j][2], wb[2]);
    OIS_PR lim[j] = S.mu * ln[j];
    // longitudinal row (couples the wheel spin)
    OIS_PR u[j] = fma2(a1[j][0], ox[j], u1);
    OIS_PR u[j] = fma2(a1[j][1], oy[j], u[j]);
    OIS_PR u[j] = fma2(a1[j][2], oz[j], u[j]);
    OIS_PR u[j] = fma2(-K.rw, spin[j], u[j]);
    OIS_PR nw[j] = clamp2(fma2(-u[j], m1[j], l1[j]), lim[j]);
    OIS_PR d1[j] = nw[j] - l1[j];
    OIS_PR l1[j] = nw[j];
    OIS_PR ox[j] = fma2(d1[j], Ja1[j][0], ox[j]);
    OIS_PR oy[j] = fma2(d1[j], Ja1[j][1], oy[j]);
    OIS_PR oz[j] = fma2(d1[j], Ja1[j][2], oz[j]);
    OIS_PR spin[j] = fma2(-K.rw_over_Is, d1[j], spin[j]);
    // lateral row
    OIS_PR u[j] = fma2(a2[j][0], ox[j], u2);
    OIS_PR u[j] = fma2(a2[j][1], oy[j], u[j]);
    OIS_PR u[j] = fma2(a2[j][2], oz[j], u[j]);
    OIS_PR nw[j] = clamp2(fma2(-u[j], m2[j], l2[j]), lim[j]);
    OIS_PR dl[j] = nw[j] - l2[j];
    OIS_PR l2[j] = nw[j];
    OIS_PR ox[j] = fma2(dl[j], Ja2[j][0], ox[j]);
    OIS_PR oy[j] = fma2(dl[j], Ja2[j][1], oy[j]);
    OIS_PR oz[j] = fma2(dl[j], Ja2[j][2], oz[j]);
    // mass-splitting average over the 4 contacts
    {
      const float Sn = sum4(dn), S1 = sum4(d1), S2 = sum4(dl);
      un = __builtin_fmaf(K.imq, Sn, un); u1 = __builtin_fmaf(K.imq, S1, u1); u2 = __builtin_fmaf(K.imq, S2, u2);
    }
    wb[0] = 0.25f * sum4(ox); wb[1] = 0.25f * sum4(oy); wb[2] = 0.25f * sum4(oz);
  }
  // v_b from its projections on the (orthonormal) contact frame; t1 has no y component
  vb[0] = __builtin_fmaf(t2x, u2, __builtin_fmaf(t1x, u1, nb[0] * un));
  vb[1] = __builtin_fmaf(t2y, u2, nb[1] * un);
  vb[2] = __builtin_fmaf(t2z, u2, __builtin_fmaf(t1z, u1, nb[2] * un));
  OIS_PR {
    st2(S.spin, j, clamp2(spin[j], (f2)(K.maxw)));
    st2(S.ln, j, ln[j]); st2(S.l1, j, l1[j]); st2(S.l2, j, l2[j]);
    st2(S.mn, j, mn[j]); st2(S.m1, j, m1[j]); st2(S.m2, j, m2[j]);
  }
  // linear velocity back to the world frame, velocity caps (origin_v18.py:65-70), integrate
  rot(R, vb, S.v);
  const float w2 = __builtin_fmaf(wb[2], wb[2], __builtin_fmaf(wb[1], wb[1], wb[0] * wb[0]));
  if (w2 > K.maxw2) {
    const float sc = K.maxw / sqrtf(w2);
    wb[0] *= sc; wb[1] *= sc; wb[2] *= sc;
  }
  S.wb[0] = wb[0]; S.wb[1] = wb[1]; S.wb[2] = wb[2];
  S.p[0] = __builtin_fmaf(K.dt, S.v[0], S.p[0]);
  S.p[1] = __builtin_fmaf(K.dt, S.v[1], S.p[1]);
  S.p[2] = __builtin_fmaf(K.dt, S.v[2], S.p[2]);
  const float qw = S.q[0], qx = S.q[1], qy = S.q[2], qz = S.q[3], h = K.half_dt;
  const float nqw = __builtin_fmaf(-h, __builtin_fmaf(qz, wb[2], __builtin_fmaf(qy, wb[1], qx * wb[0])), qw);
  const float nqx = __builtin_fmaf(h, __builtin_fmaf(-qz, wb[1], __builtin_fmaf(qy, wb[2], qw * wb[0])), qx);
  const float nqy = __builtin_fmaf(h, __builtin_fmaf(-qx, wb[2], __builtin_fmaf(qz, wb[0], qw * wb[1])), qy);
  const float nqz = __builtin_fmaf(h, __builtin_fmaf(-qy, wb[0], __builtin_fmaf(qx, wb[1], qw * wb[2])), qz);
  const float nn = __builtin_fmaf(nqz, nqz, __builtin_fmaf(nqy, nqy, __builtin_fmaf(nqx, nqx, nqw * nqw)));
  float inv = __builtin_fmaf(-0.5f, nn, 1.5f);
  if (fabsf(nn - 1.0f) > 1e-3f) inv = 1.0f / sqrtf(nn);
  S.q[0] = nqw * inv; S.q[1] = nqx * inv; S.q[2] = nqy * inv; S.q[3] = nqz * inv;
  if constexpr (TER && TMODE != kOddSubstep) footprints_store(S, cur);
#undef OIS_PR
}

// ---- LPA = 2 on packed fp32: a lane pair per arena, lane h owns the SIDE pair (wheels 2h, 2h + 1 = front, rear
// of the left / right side) in the halves of float2 operands; the chassis is replicated in both lanes and a sum
// over the four wheels is the lane's pair sum plus the partner's (one quad_perm DPP add), which is the SPEC order
// (w0 + w1) + (w2 + w3). Same operations, element for element, as substep<EXACT, 2> (ois_physics.h). For arena
// counts that leave SIMDs empty with one lane per arena (N <= 32 768 on an MI355X): the per-lane instruction
// stream is shorter (one packed pair instead of two) and the machine still holds one wave per SIMD.
OIS_DEV float sum_side(f2 c) { return side_sum<2>(c.x + c.y); }

template <bool EXACT, int NIT = 0, bool FLAT = false>
OIS_DEV void substep_pk2(const PhysK& K, int wheel0, PhysT<2>& S) {
  float R[9];
  rotmat(S.q, R);
  float vb[3], nb[3], wb[3] = {S.wb[0], S.wb[1], S.wb[2]};
  const float n[3] = {S.n[0], S.n[1], S.n[2]};
  rotT(R, S.v, vb);
  if constexpr (FLAT) { nb[0] = R[6]; nb[1] = R[7]; nb[2] = R[8]; }
  else rotT(R, n, nb);
  vb[0] = __builtin_fmaf(-K.gdt, R[6], vb[0]);
  vb[1] = __builtin_fmaf(-K.gdt, R[7], vb[1]);
  vb[2] = __builtin_fmaf(-K.gdt, R[8], vb[2]);
  f2 spin = {S.spin[0], S.spin[1]}, ln = {S.ln[0], S.ln[1]}, l1 = {S.l1[0], S.l1[1]}, l2 = {S.l2[0], S.l2[1]};
  f2 mn = {S.mn[0], S.mn[1]}, m1 = {S.m1[0], S.m1[1]}, m2 = {S.m2[0], S.m2[1]};
  const f2 tau = {S.tau[0], S.tau[1]};
  // wheel drive + joint resistance, reaction on the chassis
  const f2 jv = spin - wb[1];
  const f2 over = jv - clamp2(jv, (f2)(K.wlim));
  const f2 Tq = fma2(-K.climit, over, fma2(-K.damping, jv, tau));
  spin = fma2(K.dt_over_Is, Tq, spin);
  wb[1] = __builtin_fmaf(-K.dt_over_Iy, sum_side(Tq), wb[1]);
  // contact frame shared by the four wheels
  float s2 = __builtin_fmaf(nb[2], nb[2], nb[0] * nb[0]);
  s2 = fmaxf(s2, 1e-12f);
  const float inv_s = EXACT ? exact_rcp(sqrtf(s2)) : newton_rsqrt(S.inv_s, s2);
  S.inv_s = inv_s;
  const float dx = -(nb[0] * inv_s), dz = -(nb[2] * inv_s);
  const float t1x = -dz, t1z = dx;
  const float t2x = nb[1] * t1z;
  const float t2y = __builtin_fmaf(nb[2], t1x, -(nb[0] * t1z));
  const float t2z = -(nb[1] * t1x);
  const float np = FLAT ? S.p[2] : __builtin_fmaf(n[2], S.p[2], __builtin_fmaf(n[1], S.p[1], n[0] * S.p[0]));
  const float cz = K.cz;
  const f2 cx = {K.wx, -K.wx};                    // (front, rear) wheel of the side
  const float cy = (wheel0 & 2) ? -K.wy : K.wy;   // left side = +y (urdf:112-142)
  const f2 rx = fma2(K.rw, (f2)(dx), cx);
  const float ry = cy, rz = __builtin_fmaf(K.rw, dz, cz);
  const f2 gap = np + fma2(nb[2], (f2)(rz), fma2(nb[1], (f2)(ry), nb[0] * rx));
  const f2 bA = -(gap * K.inv_dt), bB = -(gap * K.beta_inv_dt);
  f2 b;
  b.x = __builtin_fminf(__builtin_fminf(bA.x, bB.x), K.vdep);
  b.y = __builtin_fminf(__builtin_fminf(bA.y, bB.y), K.vdep);
  f2 an[3], a1[3], a2[3], Jan[3], Ja1[3], Ja2[3];
  an[0] = (f2)(__builtin_fmaf(ry, nb[2], -(rz * nb[1])));
  an[1] = fma2((f2)(rz), (f2)(nb[0]), -(rx * nb[2]));
  an[2] = fma2(rx, (f2)(nb[1]), -(ry * nb[0]));
  a1[0] = (f2)(cy * t1z);
  a1[1] = fma2((f2)(cz), (f2)(t1x), -(cx * t1z));
  a1[2] = (f2)(-(cy * t1x));
  a2[0] = (f2)(__builtin_fmaf(ry, t2z, -(rz * t2y)));
  a2[1] = fma2((f2)(rz), (f2)(t2x), -(rx * t2z));
  a2[2] = fma2(rx, (f2)(t2y), -(ry * t2x));
  Jan[0] = K.J4x * an[0]; Jan[1] = K.J4y * an[1]; Jan[2] = K.J4z * an[2];
  Ja1[0] = K.J4x * a1[0]; Ja1[1] = K.J4y * a1[1]; Ja1[2] = K.J4z * a1[2];
  Ja2[0] = K.J4x * a2[0]; Ja2[1] = K.J4y * a2[1]; Ja2[2] = K.J4z * a2[2];
  const f2 kn = fma2(an[2], Jan[2], fma2(an[1], Jan[1], fma2(an[0], Jan[0], K.im4)));
  const f2 k1 = fma2(a1[2], Ja1[2], fma2(a1[1], Ja1[1], fma2(a1[0], Ja1[0], K.im4))) + K.rw2_over_Is;
  const f2 k2 = fma2(a2[2], Ja2[2], fma2(a2[1], Ja2[1], fma2(a2[0], Ja2[0], K.im4)));
  if (EXACT) { mn = exact_rcp2(kn); m1 = exact_rcp2(k1); m2 = exact_rcp2(k2); }
  else { mn = newton_rcp2(mn, kn); m1 = newton_rcp2(m1, k1); m2 = newton_rcp2(m2, k2); }
  float un = __builtin_fmaf(nb[2], vb[2], __builtin_fmaf(nb[1], vb[1], nb[0] * vb[0]));
  float u1 = __builtin_fmaf(t1z, vb[2], t1x * vb[0]);
  float u2 = __builtin_fmaf(t2z, vb[2], __builtin_fmaf(t2y, vb[1], t2x * vb[0]));
  // warm start
  f2 ox = fma2(l2, Ja2[0], fma2(l1, Ja1[0], fma2(ln, Jan[0], wb[0])));
  f2 oy = fma2(l2, Ja2[1], fma2(l1, Ja1[1], fma2(ln, Jan[1], wb[1])));
  f2 oz = fma2(l2, Ja2[2], fma2(l1, Ja1[2], fma2(ln, Jan[2], wb[2])));
  spin = fma2(-K.rw_over_Is, l1, spin);
  {
    const float Sn = sum_side(ln), S1 = sum_side(l1), S2 = sum_side(l2);
    un = __builtin_fmaf(K.imq, Sn, un); u1 = __builtin_fmaf(K.imq, S1, u1); u2 = __builtin_fmaf(K.imq, S2, u2);
  }
  wb[0] = 0.25f * sum_side(ox); wb[1] = 0.25f * sum_side(oy); wb[2] = 0.25f * sum_side(oz);
  const int iters = NIT > 0 ? NIT : K.iters;
#pragma unroll
  for (int it = 0; it < iters; ++it) {
    // normal row
    f2 u = fma2(an[2], wb[2], fma2(an[1], wb[1], fma2(an[0], wb[0], un)));
    f2 nw = max2(fma2(b - u, mn, ln), (f2)(0.0f));
    const f2 dn = nw - ln;
    ln = nw;
    ox = fma2(dn, Jan[0], wb[0]); oy = fma2(dn, Jan[1], wb[1]); oz = fma2(dn, Jan[2], wb[2]);
    const f2 lim = S.mu * ln;
    // longitudinal row (couples the wheel spin)
    u = fma2(-K.rw, spin, fma2(a1[2], oz, fma2(a1[1], oy, fma2(a1[0], ox, u1))));
    nw = clamp2(fma2(-u, m1, l1), lim);
    const f2 d1 = nw - l1;
    l1 = nw;
    ox = fma2(d1, Ja1[0], ox); oy = fma2(d1, Ja1[1], oy); oz = fma2(d1, Ja1[2], oz);
    spin = fma2(-K.rw_over_Is, d1, spin);
    // lateral row
    u = fma2(a2[2], oz, fma2(a2[1], oy, fma2(a2[0], ox, u2)));
    nw = clamp2(fma2(-u, m2, l2), lim);
    const f2 d2 = nw - l2;
    l2 = nw;
    ox = fma2(d2, Ja2[0], ox); oy = fma2(d2, Ja2[1], oy); oz = fma2(d2, Ja2[2], oz);
    {
      const float Sn = sum_side(dn), S1 = sum_side(d1), S2 = sum_side(d2);
      un = __builtin_fmaf(K.imq, Sn, un); u1 = __builtin_fmaf(K.imq, S1, u1); u2 = __builtin_fmaf(K.imq, S2, u2);
    }
    wb[0] = 0.25f * sum_side(ox); wb[1] = 0.25f * sum_side(oy); wb[2] = 0.25f * sum_side(oz);
  }
  vb[0] = __builtin_fmaf(t2x, u2, __builtin_fmaf(t1x, u1, nb[0] * un));
  vb[1] = __builtin_fmaf(t2y, u2, nb[1] * un);
  vb[2] = __builtin_fmaf(t2z, u2, __builtin_fmaf(t1z, u1, nb[2] * un));
  {
    const f2 sp = clamp2(spin, (f2)(K.maxw));
    S.spin[0] = sp.x; S.spin[1] = sp.y;
    S.ln[0] = ln.x; S.ln[1] = ln.y; S.l1[0] = l1.x; S.l1[1] = l1.y; S.l2[0] = l2.x; S.l2[1] = l2.y;
    S.mn[0] = mn.x; S.mn[1] = mn.y; S.m1[0] = m1.x; S.m1[1] = m1.y; S.m2[0] = m2.x; S.m2[1] = m2.y;
  }
  rot(R, vb, S.v);
  const float w2 = __builtin_fmaf(wb[2], wb[2], __builtin_fmaf(wb[1], wb[1], wb[0] * wb[0]));
  if (w2 > K.maxw2) {
    const float sc = K.maxw / sqrtf(w2);
    wb[0] *= sc; wb[1] *= sc; wb[2] *= sc;
  }
  S.wb[0] = wb[0]; S.wb[1] = wb[1]; S.wb[2] = wb[2];
  S.p[0] = __builtin_fmaf(K.dt, S.v[0], S.p[0]);
  S.p[1] = __builtin_fmaf(K.dt, S.v[1], S.p[1]);
  S.p[2] = __builtin_fmaf(K.dt, S.v[2], S.p[2]);
  const float qw = S.q[0], qx = S.q[1], qy = S.q[2], qz = S.q[3], h = K.half_dt;
  const float nqw = __builtin_fmaf(-h, __builtin_fmaf(qz, wb[2], __builtin_fmaf(qy, wb[1], qx * wb[0])), qw);
  const float nqx = __builtin_fmaf(h, __builtin_fmaf(-qz, wb[1], __builtin_fmaf(qy, wb[2], qw * wb[0])), qx);
  const float nqy = __builtin_fmaf(h, __builtin_fmaf(-qx, wb[2], __builtin_fmaf(qz, wb[0], qw * wb[1])), qy);
  const float nqz = __builtin_fmaf(h, __builtin_fmaf(-qy, wb[0], __builtin_fmaf(qx, wb[1], qw * wb[2])), qz);
  const float nn = __builtin_fmaf(nqz, nqz, __builtin_fmaf(nqy, nqy, __builtin_fmaf(nqx, nqx, nqw * nqw)));
  float inv = __builtin_fmaf(-0.5f, nn, 1.5f);
  if (fabsf(nn - 1.0f) > 1e-3f) inv = 1.0f / sqrtf(nn);
  S.q[0] = nqw * inv; S.q[1] = nqx * inv; S.q[2] = nqy * inv; S.q[3] = nqz * inv;
}

}  // namespace ois
