// ois_physics.h — one physics substep of one arena (docs/SPEC.md §3).
//
// Rigid chassis (6 DoF) + 4 wheel spins, 4 wheel-ground contact points on a plane, each with
// a normal row and two Coulomb (box) friction rows; the longitudinal row couples the wheel
// spin. Solver: mass-splitting Jacobi across the 4 contacts (each contact solves against a
// private copy of the chassis velocity with 4x the inverse mass, the copies are averaged),
// Gauss-Seidel across the 3 rows of a contact, 4 iterations, warm-started from the previous
// substep's impulses.
//
// Written once for LPA = 1 (one lane owns all 4 contacts) and LPA = 2 (two lanes own the
// left / right pair of contacts; cross-side sums are one DPP add). The contacts of a lane are
// walked in LOCK-STEP (one dot-product / update stage for all of them before the next stage)
// so that the in-order SIMD has independent instructions in flight.
// Mirrors oracle/ois_oracle.c substep() operation for operation.
#pragma once
#include "ois_device.h"

namespace ois {

template <int LPA>
struct PhysT {
  static constexpr int WPL = 4 / LPA;
  float p[3], q[4], v[3];  // COM position, orientation (w,x,y,z), WORLD linear velocity
  float wb[3];             // BODY-frame angular velocity (constant between substeps, SPEC §3.2)
  float n[3], mu;          // ground plane normal (world), friction coefficient
  float tau[WPL];          // clipped drive torques of this lane's wheels
  float spin[WPL];         // absolute wheel spins about the body y axis
  float ln[WPL], l1[WPL], l2[WPL];         // accumulated normal / longitudinal / lateral impulses
  float mn[WPL], m1[WPL], m2[WPL], inv_s;  // reciprocal cache (SPEC §3.6), valid inside one policy step
  float to[3];             // origin of the arena's terrain tile (only read by the TER instantiations)
  // wheel footprints sampled at the start of the previous substep (TER only, SPEC §3.8)
  float fwx[4], fwy[4], fh[4], fr00, fr10, fr01, fr11;
  int fpx[4], fpy[4];      // field cell of each stored footprint (gather cache, see terrain_footprints)
  float pn[3], poff[4];    // footprint stride 2 only: the plane an odd substep fitted, kept by the even substep after it
};

// 1.0f / d, correctly rounded, for finite d with 2^-40 <= |d| < 2^40: the refinement chain of the
// compiler's own fdiv expansion (v_rcp_f32 + 6 fma) without its v_div_scale / v_div_fmas /
// v_div_fixup, which only act at the exponent extremes: 7 instructions instead of 12, and the
// chain packs (ois_physics_pk.h). Bit-identical to the IEEE quotient the oracle computes on every
// float of that range (tools/exact_rcp_check.hip: 671 M values, exhaustive; tests/test_gpu_parity.py).
// The contact-row diagonals it is used on are >= 1 / (4 m) > 0 for any finite geometry; NaN stays NaN.
OIS_DEV float exact_rcp(float d) {
  const float r0 = __builtin_amdgcn_rcpf(d);
  const float e = __builtin_fmaf(-d, r0, 1.0f);
  const float r1 = __builtin_fmaf(e, r0, r0);
  const float g = __builtin_fmaf(-d, r1, 1.0f);
  const float q = __builtin_fmaf(g, r1, r1);
  const float h = __builtin_fmaf(-d, q, 1.0f);
  return __builtin_fmaf(h, r1, q);
}

// One Newton step from the previous substep's value: mul/fma only, bit-reproducible.
OIS_DEV float newton_rcp(float m, float k) {  // m ~ 1/k -> m (2 - k m)
  const float e = __builtin_fmaf(-k, m, 1.0f);
  return __builtin_fmaf(m, e, m);
}
OIS_DEV float newton_rsqrt(float r, float s2) {  // r ~ 1/sqrt(s2) -> r (1.5 - 0.5 s2 r^2)
  const float hr = (0.5f * s2) * r;
  const float e = __builtin_fmaf(-hr, r, 1.5f);
  return r * e;
}

// ---- height-field terrain (docs/SPEC.md §3.8), mirrors oracle/ois_oracle.c terrain_height / terrain_plane ----
// ground height at world (x, y): get_heights' sampling rule (KAMMA.py:882-893) with the division
// by horizontal_scale replaced by a multiplication; positions are clamped before the int
// conversion, so no index can leave the field whatever the state (NaN clamps to 0)
OIS_DEV void terrain_cell(const DevTerrain& T, float x, float y, int& px, int& py) {
  float fx = (x + T.border) * T.inv_hs, fy = (y + T.border) * T.inv_hs;
  // clamp as one v_med3_f32 (a NaN comes out as 0 either way: med3 with a NaN operand is min3,
  // which skips it, and the int conversion of a NaN is 0 — no index can leave the field)
  fx = __builtin_amdgcn_fmed3f(fx, 0.0f, (float)(T.rows - 2));
  fy = __builtin_amdgcn_fmed3f(fy, 0.0f, (float)(T.cols - 2));
  px = (int)fx; py = (int)fy;
}
OIS_DEV float terrain_cell_height(const DevTerrain& T, int px, int py) {
  // m(x, y) = min(h[x][y], h[x+1][y+1]), built once at ois_terrain_attach: one gather per sample. Stored in
  // tiles of 8 x 8 cells = 128 B (one cache line): the 140 scan points of a robot are 5 cells apart and rotated
  // by its yaw, so in a row-major field nearly every point is on a line of its own whatever the heading; a
  // tile holds 2-3 of them (height scan at 65 536 arenas 53.9 -> 41.8 us, tools/height_scan_probe.py)
  // 32-bit byte offset (ois_terrain_attach refuses a field of 2^31 bytes): the gather takes the base from SGPRs
  const unsigned ux = (unsigned)px, uy = (unsigned)py;
  const unsigned off = (((ux >> 3) * (unsigned)T.tiles_per_row + (uy >> 3)) << 7) + (((ux & 7u) << 4) | ((uy & 7u) << 1));
  return (float)*reinterpret_cast<const int16_t*>(reinterpret_cast<const char*>(T.m) + off) * T.vs;
}
// footprints of the 4 wheels at the current pose: world xy under the wheel centres and the field
// height there (4 gathers of the min-field, issued here, consumed one substep later). A wheel moves at most ~2 cm
// per substep, so its cell is often the one sampled a substep earlier: those lanes reuse the
// stored height (same value by construction, the tile origin is fixed inside a policy step)
// instead of gathering again. All waves reach this point together, so the gathers of a substep
// are one burst of line requests against L2; thinning it is what shortens the substep.
struct Footprints {
  float wx[4], wy[4], h[4], r00, r10, r01, r11;
  int px[4], py[4];
};
template <bool FRESH, int LPA>
OIS_DEV void terrain_footprints(const PhysK& K, const DevTerrain& T, const PhysT<LPA>& S, const float* R, Footprints& F) {
  const float* origin = S.to;
  const float* p = S.p;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float cx = (k & 1) ? -K.wx : K.wx, cy = (k & 2) ? -K.wy : K.wy;
    F.wx[k] = __builtin_fmaf(R[2], K.cz, __builtin_fmaf(R[1], cy, __builtin_fmaf(R[0], cx, p[0])));
    F.wy[k] = __builtin_fmaf(R[5], K.cz, __builtin_fmaf(R[4], cy, __builtin_fmaf(R[3], cx, p[1])));
    terrain_cell(T, F.wx[k] + origin[0], F.wy[k] + origin[1], F.px[k], F.py[k]);
    // unchanged cell: the lane re-reads cell (0, 0) instead (one shared line, an L1 hit) and keeps
    // the stored height; the loads stay unconditional, so nothing waits for them before the latch
    // (the mask goes through an empty asm so that the compiler cannot turn the blend back into a
    // branch around the loads, which would put their latency back in front of the solve)
    int m = (!FRESH && F.px[k] == S.fpx[k] && F.py[k] == S.fpy[k]) ? -1 : 0;
    asm volatile("" : "+v"(m));
    const float hh = terrain_cell_height(T, F.px[k] & ~m, F.py[k] & ~m) - origin[2];
    F.h[k] = __int_as_float((__float_as_int(S.fh[k]) & m) | (__float_as_int(hh) & ~m));
  }
  F.r00 = R[0]; F.r10 = R[3]; F.r01 = R[1]; F.r11 = R[4];
}
template <int LPA>
OIS_DEV void footprints_store(PhysT<LPA>& S, const Footprints& F) {
#pragma unroll
  for (int k = 0; k < 4; ++k) { S.fwx[k] = F.wx[k]; S.fwy[k] = F.wy[k]; S.fh[k] = F.h[k]; S.fpx[k] = F.px[k]; S.fpy[k] = F.py[k]; }
  S.fr00 = F.r00; S.fr10 = F.r10; S.fr01 = F.r01; S.fr11 = F.r11;
}
// plane through the stored footprints: unit normal n (world), off[k] = n . footprint_k
template <int LPA>
OIS_DEV void terrain_plane(const PhysK& K, const PhysT<LPA>& S, float* n, float* off) {
  const float* h = S.fh;
  const float g1 = ((h[0] + h[2]) - (h[1] + h[3])) * K.inv4wx;  // front - rear (order fl, rl, fr, rr)
  const float g2 = ((h[0] + h[1]) - (h[2] + h[3])) * K.inv4wy;  // left - right
  const float nx = __builtin_fmaf(S.fr10, g2, -(g1 * S.fr11));
  const float ny = __builtin_fmaf(g1, S.fr01, -(S.fr00 * g2));
  const float nz = __builtin_fmaf(S.fr00, S.fr11, -(S.fr10 * S.fr01));
  const float nn = fmaxf(__builtin_fmaf(nz, nz, __builtin_fmaf(ny, ny, nx * nx)), 1e-12f);
  float inv = exact_rcp(sqrtf(nn));  // 1e-12 <= nn: inside exact_rcp's range
  if (nz < 0.0f) inv = -inv;  // rolled over (body z points down): the ground's normal still points up
  n[0] = nx * inv; n[1] = ny * inv; n[2] = nz * inv;
#pragma unroll
  for (int k = 0; k < 4; ++k) off[k] = __builtin_fmaf(n[2], h[k], __builtin_fmaf(n[1], S.fwy[k], n[0] * S.fwx[k]));
}

// 0.25 * ((c0 + c1) + (c2 + c3)) over the per-contact copies of one velocity component
template <int LPA>
OIS_DEV float average4(const float (&c)[4 / LPA]) { return 0.25f * sum_wheels<LPA>(c); }

// EXACT: first substep of a policy step (IEEE divisions); otherwise the reciprocals are
// refreshed from the cache. wheel0 = index of this lane's first wheel (0, or 2 for the
// right-side lane when LPA = 2).
// NIT: solver iterations known at compile time (the loop is fully unrolled: no taken branch,
// scheduling across iteration boundaries) or 0 = K.iters at run time
// TMODE (height field with ois_terrain.footprint_stride = 2, SPEC 3.8; 0 = every substep samples): kOddSubstep fits the plane
// to the stored sample and samples nothing; kEvenSubstep samples (for the substep after it) and keeps the plane of the substep
// before it.
constexpr int kEverySubstep = 0, kOddSubstep = 1, kEvenSubstep = 2;
template <bool EXACT, int LPA, bool TER = false, int NIT = 0, int TMODE = kEverySubstep>
OIS_DEV void substep(const PhysK& K, const DevTerrain& T, int wheel0, PhysT<LPA>& S) {
  static_assert(TMODE == kEverySubstep || (TER && !EXACT), "footprint stride modes: later substeps on a height field");
  static_assert(!TER || LPA == 1, "the terrain contact needs all 4 wheels in one lane");
  constexpr int W = 4 / LPA;
#define OIS_WH _Pragma("unroll") for (int k = 0; k < W; ++k)
  float R[9];
  rotmat(S.q, R);
  float vb[3], nb[3], wb[3] = {S.wb[0], S.wb[1], S.wb[2]};
  float n[3] = {S.n[0], S.n[1], S.n[2]}, off[4] = {0.0f, 0.0f, 0.0f, 0.0f};
  Footprints cur;
  if constexpr (TER) {
    // fit the plane to the footprints sampled one substep earlier (the first substep of a policy
    // step samples its own): this substep's 8 gathers stay in flight during the whole solve
    if constexpr (TMODE != kOddSubstep) terrain_footprints<EXACT>(K, T, S, R, cur);
    if (EXACT) footprints_store(S, cur);
    if constexpr (TMODE == kEvenSubstep) {
#pragma unroll
      for (int k = 0; k < 3; ++k) n[k] = S.pn[k];
#pragma unroll
      for (int k = 0; k < 4; ++k) off[k] = S.poff[k];
    } else {
      terrain_plane(K, S, n, off);
      if constexpr (TMODE == kOddSubstep) {
#pragma unroll
        for (int k = 0; k < 3; ++k) S.pn[k] = n[k];
#pragma unroll
        for (int k = 0; k < 4; ++k) S.poff[k] = off[k];
      }
    }
  }
  rotT(R, S.v, vb);
  rotT(R, n, nb);
  // gravity (world -z) in the body frame
  vb[0] = __builtin_fmaf(-K.gdt, R[6], vb[0]);
  vb[1] = __builtin_fmaf(-K.gdt, R[7], vb[1]);
  vb[2] = __builtin_fmaf(-K.gdt, R[8], vb[2]);
  // wheel drive + viscous joint damping, reaction on the chassis
  float Tq[W];
  OIS_WH {
    const float jv = S.spin[k] - wb[1];
    const float over = jv - __builtin_amdgcn_fmed3f(jv, -K.wlim, K.wlim);  // signed excess over the joint velocity limit
    Tq[k] = __builtin_fmaf(-K.climit, over, __builtin_fmaf(-K.damping, jv, S.tau[k]));
    S.spin[k] = __builtin_fmaf(K.dt_over_Is, Tq[k], S.spin[k]);
  }
  const float Tsum = sum_wheels<LPA>(Tq);
  wb[1] = __builtin_fmaf(-K.dt_over_Iy, Tsum, wb[1]);
  // contact frame shared by the four wheels (common axle direction = body y)
  float s2 = __builtin_fmaf(nb[2], nb[2], nb[0] * nb[0]);
  s2 = fmaxf(s2, 1e-12f);
  // on a height field exact reciprocals in EVERY substep, the plane-keeping ones of footprint stride 2 included: a robot tumbling
  // over the field passes through poses (on its side, n_b along the axle) where the contact points move by a wheel radius within
  // a substep, and a Newton refresh then returns finite garbage (round 4: 1 000 m/s launches, 1 arena in 10^4 per 10 steps)
  constexpr bool REX = EXACT || TER;
  const float inv_s = REX ? exact_rcp(sqrtf(s2)) : newton_rsqrt(S.inv_s, s2);  // s2 >= 1e-12
  S.inv_s = inv_s;
  const float dx = -(nb[0] * inv_s), dz = -(nb[2] * inv_s);
  const float t1x = -dz, t1z = dx;
  const float t2x = nb[1] * t1z;
  const float t2y = __builtin_fmaf(nb[2], t1x, -(nb[0] * t1z));
  const float t2z = -(nb[1] * t1x);
  const float np = __builtin_fmaf(n[2], S.p[2], __builtin_fmaf(n[1], S.p[1], n[0] * S.p[0]));
  const float cz = K.cz;
  float b[W], an[W][3], a1[W][3], a2[W][3], Jan[W][3], Ja1[W][3], Ja2[W][3];
  float (&mn)[W] = S.mn, (&m1)[W] = S.m1, (&m2)[W] = S.m2;
  OIS_WH {
    // wheel centres relative to the COM, env order fl, rl, fr, rr (urdf:112-142: left = +y, front = +x)
    const int gk = wheel0 + k;
    const float cx = (k & 1) ? -K.wx : K.wx;  // wheel0 is even: parity of gk = parity of k
    const float cy = (gk & 2) ? -K.wy : K.wy;
    const float rx = __builtin_fmaf(K.rw, dx, cx), ry = cy, rz = __builtin_fmaf(K.rw, dz, cz);
    const float gap = (TER ? np - off[k] : np) + __builtin_fmaf(nb[2], rz, __builtin_fmaf(nb[1], ry, nb[0] * rx));
    b[k] = __builtin_fminf(__builtin_fminf(-(gap * K.inv_dt), -(gap * K.beta_inv_dt)), K.vdep);  // = gap >= 0 ? -gap/dt : min(-beta gap/dt, vdep) for 0 < beta < 1
    an[k][0] = __builtin_fmaf(ry, nb[2], -(rz * nb[1]));
    an[k][1] = __builtin_fmaf(rz, nb[0], -(rx * nb[2]));
    an[k][2] = __builtin_fmaf(rx, nb[1], -(ry * nb[0]));
    a1[k][0] = cy * t1z;
    a1[k][1] = __builtin_fmaf(cz, t1x, -(cx * t1z));
    a1[k][2] = -(cy * t1x);
    a2[k][0] = __builtin_fmaf(ry, t2z, -(rz * t2y));
    a2[k][1] = __builtin_fmaf(rz, t2x, -(rx * t2z));
    a2[k][2] = __builtin_fmaf(rx, t2y, -(ry * t2x));
    Jan[k][0] = K.J4x * an[k][0]; Jan[k][1] = K.J4y * an[k][1]; Jan[k][2] = K.J4z * an[k][2];
    Ja1[k][0] = K.J4x * a1[k][0]; Ja1[k][1] = K.J4y * a1[k][1]; Ja1[k][2] = K.J4z * a1[k][2];
    Ja2[k][0] = K.J4x * a2[k][0]; Ja2[k][1] = K.J4y * a2[k][1]; Ja2[k][2] = K.J4z * a2[k][2];
    const float kn = __builtin_fmaf(an[k][2], Jan[k][2], __builtin_fmaf(an[k][1], Jan[k][1], __builtin_fmaf(an[k][0], Jan[k][0], K.im4)));
    const float k1 = __builtin_fmaf(a1[k][2], Ja1[k][2], __builtin_fmaf(a1[k][1], Ja1[k][1], __builtin_fmaf(a1[k][0], Ja1[k][0], K.im4))) + K.rw2_over_Is;
    const float k2 = __builtin_fmaf(a2[k][2], Ja2[k][2], __builtin_fmaf(a2[k][1], Ja2[k][1], __builtin_fmaf(a2[k][0], Ja2[k][0], K.im4)));
    mn[k] = REX ? exact_rcp(kn) : newton_rcp(mn[k], kn);
    m1[k] = REX ? exact_rcp(k1) : newton_rcp(m1[k], k1);
    m2[k] = REX ? exact_rcp(k2) : newton_rcp(m2[k], k2);
  }
  // Chassis linear velocity: the contact frame (n, t1, t2) is orthonormal and the split inverse
  // mass is a scalar, so an impulse along one row changes only that row's own component of the
  // linear velocity: the solve carries the three projections (un, u1, u2) of v_b on the frame; a
  // row reads its component, the averaging pass adds (1/m) * (sum of the pass's impulses of that
  // row), and v_b is rebuilt from the frame once, after the last pass (docs/SPEC.md 3.5).
  float un = __builtin_fmaf(nb[2], vb[2], __builtin_fmaf(nb[1], vb[1], nb[0] * vb[0]));
  float u1 = __builtin_fmaf(t1z, vb[2], t1x * vb[0]);
  float u2 = __builtin_fmaf(t2z, vb[2], __builtin_fmaf(t2y, vb[1], t2x * vb[0]));
  float ox[W], oy[W], oz[W];
  // warm start: one mass-split averaging pass with the cached impulses
  OIS_WH ox[k] = __builtin_fmaf(S.l2[k], Ja2[k][0], __builtin_fmaf(S.l1[k], Ja1[k][0], __builtin_fmaf(S.ln[k], Jan[k][0], wb[0])));
  OIS_WH oy[k] = __builtin_fmaf(S.l2[k], Ja2[k][1], __builtin_fmaf(S.l1[k], Ja1[k][1], __builtin_fmaf(S.ln[k], Jan[k][1], wb[1])));
  OIS_WH oz[k] = __builtin_fmaf(S.l2[k], Ja2[k][2], __builtin_fmaf(S.l1[k], Ja1[k][2], __builtin_fmaf(S.ln[k], Jan[k][2], wb[2])));
  OIS_WH S.spin[k] = __builtin_fmaf(-K.rw_over_Is, S.l1[k], S.spin[k]);
  {
    const float Sn = sum_wheels<LPA>(S.ln), S1 = sum_wheels<LPA>(S.l1), S2 = sum_wheels<LPA>(S.l2);
    un = __builtin_fmaf(K.imq, Sn, un); u1 = __builtin_fmaf(K.imq, S1, u1); u2 = __builtin_fmaf(K.imq, S2, u2);
  }
  wb[0] = average4<LPA>(ox); wb[1] = average4<LPA>(oy); wb[2] = average4<LPA>(oz);
  const int iters = NIT > 0 ? NIT : K.iters;
#pragma unroll
  for (int it = 0; it < iters; ++it) {
    float u[W], dl[W], nw[W], lim[W], dn[W], d1[W];
    // normal row
    OIS_WH u[k] = __builtin_fmaf(an[k][0], wb[0], un);
    OIS_WH u[k] = __builtin_fmaf(an[k][1], wb[1], u[k]);
    OIS_WH u[k] = __builtin_fmaf(an[k][2], wb[2], u[k]);
    OIS_WH nw[k] = fmaxf(__builtin_fmaf(b[k] - u[k], mn[k], S.ln[k]), 0.0f);
    OIS_WH dn[k] = nw[k] - S.ln[k];
    OIS_WH S.ln[k] = nw[k];
    OIS_WH ox[k] = __builtin_fmaf(dn[k], Jan[k][0], wb[0]);
    OIS_WH oy[k] = __builtin_fmaf(dn[k], Jan[k][1], wb[1]);
    OIS_WH oz[k] = __builtin_fmaf(dn[k], Jan[k][2], wb[2]);
    OIS_WH lim[k] = S.mu * S.ln[k];
    // longitudinal row (couples the wheel spin)
    OIS_WH u[k] = __builtin_fmaf(a1[k][0], ox[k], u1);
    OIS_WH u[k] = __builtin_fmaf(a1[k][1], oy[k], u[k]);
    OIS_WH u[k] = __builtin_fmaf(a1[k][2], oz[k], u[k]);
    OIS_WH u[k] = __builtin_fmaf(-K.rw, S.spin[k], u[k]);
    OIS_WH nw[k] = fminf(fmaxf(__builtin_fmaf(-u[k], m1[k], S.l1[k]), -lim[k]), lim[k]);
    OIS_WH d1[k] = nw[k] - S.l1[k];
    OIS_WH S.l1[k] = nw[k];
    OIS_WH ox[k] = __builtin_fmaf(d1[k], Ja1[k][0], ox[k]);
    OIS_WH oy[k] = __builtin_fmaf(d1[k], Ja1[k][1], oy[k]);
    OIS_WH oz[k] = __builtin_fmaf(d1[k], Ja1[k][2], oz[k]);
    OIS_WH S.spin[k] = __builtin_fmaf(-K.rw_over_Is, d1[k], S.spin[k]);
    // lateral row
    OIS_WH u[k] = __builtin_fmaf(a2[k][0], ox[k], u2);
    OIS_WH u[k] = __builtin_fmaf(a2[k][1], oy[k], u[k]);
    OIS_WH u[k] = __builtin_fmaf(a2[k][2], oz[k], u[k]);
    OIS_WH nw[k] = fminf(fmaxf(__builtin_fmaf(-u[k], m2[k], S.l2[k]), -lim[k]), lim[k]);
    OIS_WH dl[k] = nw[k] - S.l2[k];
    OIS_WH S.l2[k] = nw[k];
    OIS_WH ox[k] = __builtin_fmaf(dl[k], Ja2[k][0], ox[k]);
    OIS_WH oy[k] = __builtin_fmaf(dl[k], Ja2[k][1], oy[k]);
    OIS_WH oz[k] = __builtin_fmaf(dl[k], Ja2[k][2], oz[k]);
    // mass-splitting average over the 4 contacts
    {
      const float Sn = sum_wheels<LPA>(dn), S1 = sum_wheels<LPA>(d1), S2 = sum_wheels<LPA>(dl);
      un = __builtin_fmaf(K.imq, Sn, un); u1 = __builtin_fmaf(K.imq, S1, u1); u2 = __builtin_fmaf(K.imq, S2, u2);
    }
    wb[0] = average4<LPA>(ox); wb[1] = average4<LPA>(oy); wb[2] = average4<LPA>(oz);
  }
  // v_b from its projections on the (orthonormal) contact frame; t1 has no y component
  vb[0] = __builtin_fmaf(t2x, u2, __builtin_fmaf(t1x, u1, nb[0] * un));
  vb[1] = __builtin_fmaf(t2y, u2, nb[1] * un);
  vb[2] = __builtin_fmaf(t2z, u2, __builtin_fmaf(t1z, u1, nb[2] * un));
  // linear velocity back to the world frame, velocity caps (origin_v18.py:65-70), integrate
  rot(R, vb, S.v);
  const float w2 = __builtin_fmaf(wb[2], wb[2], __builtin_fmaf(wb[1], wb[1], wb[0] * wb[0]));
  if (w2 > K.maxw2) {
    const float sc = K.maxw / sqrtf(w2);
    wb[0] *= sc; wb[1] *= sc; wb[2] *= sc;
  }
  S.wb[0] = wb[0]; S.wb[1] = wb[1]; S.wb[2] = wb[2];
  OIS_WH S.spin[k] = fminf(fmaxf(S.spin[k], -K.maxw), K.maxw);
  S.p[0] = __builtin_fmaf(K.dt, S.v[0], S.p[0]);
  S.p[1] = __builtin_fmaf(K.dt, S.v[1], S.p[1]);
  S.p[2] = __builtin_fmaf(K.dt, S.v[2], S.p[2]);
  // q += dt/2 * q (x) (0, w_b), then renormalise
  const float qw = S.q[0], qx = S.q[1], qy = S.q[2], qz = S.q[3], h = K.half_dt;
  const float nqw = __builtin_fmaf(-h, __builtin_fmaf(qz, wb[2], __builtin_fmaf(qy, wb[1], qx * wb[0])), qw);
  const float nqx = __builtin_fmaf(h, __builtin_fmaf(-qz, wb[1], __builtin_fmaf(qy, wb[2], qw * wb[0])), qx);
  const float nqy = __builtin_fmaf(h, __builtin_fmaf(-qx, wb[2], __builtin_fmaf(qz, wb[0], qw * wb[1])), qy);
  const float nqz = __builtin_fmaf(h, __builtin_fmaf(-qy, wb[0], __builtin_fmaf(qx, wb[1], qw * wb[2])), qz);
  const float nn = __builtin_fmaf(nqz, nqz, __builtin_fmaf(nqy, nqy, __builtin_fmaf(nqx, nqx, nqw * nqw)));
  // |q|^2 = 1 + O((w dt)^2): one Newton step of 1/sqrt is exact to float precision; the exact
  // form only when the norm has drifted (extreme spin)
  float inv = __builtin_fmaf(-0.5f, nn, 1.5f);
  if (fabsf(nn - 1.0f) > 1e-3f) inv = 1.0f / sqrtf(nn);
  S.q[0] = nqw * inv; S.q[1] = nqx * inv; S.q[2] = nqy * inv; S.q[3] = nqz * inv;
  if constexpr (TER && TMODE != kOddSubstep) footprints_store(S, cur);
#undef OIS_WH
}

}  // namespace ois
