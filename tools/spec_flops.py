#!/usr/bin/env python3
"""Algorithmic floating-point work of one physics substep and one policy step, counted from docs/SPEC.md section 3
(the MODEL's operations — not any kernel's instructions): the figure behind `roofline.flop_algorithmic_per_env_step`
in bench.py's line. Conventions: add / sub / mul / min / max / compare-select = 1, fma = 2, a reciprocal or inverse square
root refresh = its Newton step's operations (SPEC 3.6), structural zeros the SPEC itself states (`t1.y = 0`, `d.y = 0`,
`t1 = (-d.z, 0, d.x)`) are not multiplied out. One row per SPEC paragraph; (per item, items) -> flop.

    python tools/spec_flops.py          prints the table of docs/SPEC.md section 3.11
"""
SOLVER_ITERATIONS = 4
SUBSTEPS_PER_POLICY_STEP = 20   # decimation 10 (KAMMA.py:646) + controlFrequencyInv 10 (KAMMA.py:43)

# (SPEC paragraph, what, flop per item, items per substep)
SUBSTEP = [
    ("3.2", "R(q): 9 products, 3 diagonal entries 1 - 2(a + b) (add + fma), 6 off-diagonal 2(a +- b) (add + mul)", 9 + 3 * 3 + 6 * 2, 1),
    ("3.2", "v_b = R^T v, n_b = R^T n: two 3x3 products (mul + 2 fma per component)", 15, 2),
    ("3.3", "gravity v_b -= g dt R3^T: 3 fma", 6, 1),
    ("3.3", "joint: w = Omega - w_y; clamp(w, +-w_lim) (2); w - clamp; T = tau - c1 w - c2 over (2 fma); Omega += dt/Is T (fma)", 1 + 2 + 1 + 4 + 2, 4),
    ("3.3", "reaction: sum of 4 torques (3 add), w_y -= dt/Iyy sum (fma)", 5, 1),
    ("3.4", "contact frame: s2 = nx^2 + nz^2 (mul + fma); 1/s Newton r(1.5 - 0.5 s2 r^2) (5); d = -(nx, 0, nz) r (2); t1 = (-d.z, 0, d.x) (0); "
            "t2 = n_b x t1 with t1.y = 0 (4 mul + 1 sub); n.p (mul + 2 fma)", 3 + 5 + 2 + 0 + 5 + 5, 1),
    ("3.4", "per wheel: r = c + r_w d (2 fma); gap = n.p + n_b.r (3 fma); a_n = r x n_b (9); a_1 = c x t1, t1.y = 0 (5); a_2 = r x t2 (9); "
            "(4 I^-1) a for 3 rows (9 mul); k = 4/m [+ r_w^2/Is] + a.(4 I^-1 a) for 3 rows (3 fma each); m <- m + m(1 - k m) for 3 rows (2 fma each); "
            "bias min3(-gap/dt, -beta gap/dt, v_dep) (2 mul + 2 min)", 4 + 6 + 9 + 5 + 9 + 9 + 18 + 12 + 4, 4),
    ("3.5", "projections un = n_b.v_b (5), u1 = t1.v_b with t1.y = 0 (3), u2 = t2.v_b (5)", 13, 1),
    ("3.5", "warm start, per contact: o = w_b + sum of 3 cached impulses x (4 I^-1 a) (9 fma); Omega -= r_w/Is lambda_1 (fma)", 18 + 2, 4),
    ("3.5", "warm start, averaging: S_n, S_1, S_2 (3 x 3 add); u += S/m (3 fma); w_b = 0.25 ((o0 + o1) + (o2 + o3)) (3 x (3 add + mul))", 9 + 6 + 12, 1),
    ("3.5", "iteration, per contact: normal row u = un + a_n.o (3 fma), b - u, fma, max, delta, o += delta (4 I^-1 a_n) (3 fma) = 17; "
            "longitudinal u = u1 + a_1.o - r_w Omega (4 fma), mu lambda_n', fma, clamp (2), delta, o update (3 fma), Omega update (fma) = 22; "
            "lateral (3 fma), fma, clamp (2), delta, o update (3 fma) = 17", 17 + 22 + 17, 4 * SOLVER_ITERATIONS),
    ("3.5", "iteration, averaging (as in the warm start)", 9 + 6 + 12, SOLVER_ITERATIONS),
    ("3.5", "v_b = n_b un + t1 u1 + t2 u2 (x: mul + 2 fma, y: mul + fma, z: mul + 2 fma)", 5 + 3 + 5, 1),
    ("3.7", "v = R v_b (15); cap of norm(w_b): w.w (5), rsqrt, mul, min, 3 mul (11); 4 spin caps (2 each); p += dt v (3 fma); "
            "q (x) (0, w_b): 4 x (mul + 2 fma); q += dt/2 (...) (4 fma); q.q (mul + 3 fma), 1.5 - 0.5 q.q (fma), 4 mul", 15 + 11 + 8 + 6 + 20 + 8 + 13, 1),
]

# the policy step's own arithmetic (SPEC section 2; once per 20 substeps)
POLICY_STEP = [
    ("2.2", "criteria controller: 2 differences, 2 x (mul, div, mul) + 2 (KAMMA.py:609-619)", 2 + 6 + 6 + 2, 1),
    ("2.3-2.4", "gamma (convert, div, sub, max), switch / blend per wheel (3), clip per wheel (2), guiding norm (4 sub, 4 mul, 3 add, sqrt)", 4 + 12 + 8 + 12, 1),
    ("2.5", "wheel spins <-> joint velocities around the substeps (8), body <-> world angular velocity (2 x 15)", 38, 1),
    ("2.6-2.8", "is_done (quat_apply z: 9; 4 bound compares), metrics (2 acc: 4; r1 4, r2 3, r3 7, dense 5, sparse 3, observed 1, blend 4), 8 sums", 13 + 31 + 8, 1),
    ("2.9", "projection tail: quat_apply (18), norm (6), dot (5)", 29, 1),
]


def flop_per_substep():
    return sum(f * n for _, _, f, n in SUBSTEP)


def flop_per_env_step():
    return SUBSTEPS_PER_POLICY_STEP * flop_per_substep() + sum(f * n for _, _, f, n in POLICY_STEP)


if __name__ == "__main__":
    print("| SPEC | operations | flop | x | total |\n|---|---|---|---|---|")
    for sec, what, f, n in SUBSTEP:
        print(f"| {sec} | {what} | {f} | {n} | {f * n} |")
    print(f"| | **one substep** | | | **{flop_per_substep()}** |")
    for sec, what, f, n in POLICY_STEP:
        print(f"| {sec} | {what} | {f} | {n} | {f * n} |")
    print(f"| | **one env-step** = {SUBSTEPS_PER_POLICY_STEP} substeps + the policy step's own | | | **{flop_per_env_step()}** |")
