"""Pins the CPU oracle's reference-followed math against the golden vectors of
tests/golden/refmath_*.npz (torch-CPU transcription of the cited KAMMA.py lines, see
tests/golden/make_golden.py). Element-wise float32 chains must match EXACTLY; reductions
over the 4 wheels (torch.norm / torch.sum, whose summation order is implementation defined)
are compared at 1e-6 relative."""
import ctypes as C
import os

import numpy as np
import pytest

from options_in_simulation_amd import _abi, build_config
from oracle.binding import fptr, iptr, u8ptr

G = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def step():
    return dict(np.load(os.path.join(G, "refmath_step.npz")))


def f32(x):
    return np.ascontiguousarray(x, dtype=np.float32)


def test_criteria_controller_exact(oracle_lib, step):
    n = len(step["v_delta"])
    out = [np.zeros(n, np.float32) for _ in range(4)]
    oracle_lib.orc_criteria(n, fptr(f32(step["commands"][:, 0])), fptr(f32(step["commands"][:, 2])),
                            fptr(f32(step["v_forward_projected"])), fptr(f32(step["base_velocities"][:, 5])),
                            0.035, 0.665, 25.0, 1.05, np.float32(0.1), *[fptr(o) for o in out])
    for got, key in zip(out, ("v_delta", "omega_delta", "ac_left", "ac_right")):
        assert np.array_equal(got, step[key]), key  # KAMMA.py:604-624


def test_gamma_schedule_exact(oracle_lib, step):
    n = len(step["gamma"])
    g = np.zeros(n, np.float32)
    oracle_lib.orc_gamma(n, iptr(np.ascontiguousarray(step["sim_steps"])), 400000.0, fptr(g))
    assert np.array_equal(g, step["gamma"])  # KAMMA.py:627-628
    assert g.min() == 0.0 and g.max() <= 1.0  # the clamp at 0 is exercised (sim_steps up to 600k)


@pytest.mark.parametrize("mode,prefix", [(_abi.MIX_SWITCH, "switch"), (_abi.MIX_BLEND, "blend")])
def test_action_mixing_clip_guiding(oracle_lib, step, mode, prefix):
    n = len(step["gamma"])
    tq, wt, gd = np.zeros((n, 4), np.float32), np.zeros((n, 4), np.float32), np.zeros(n, np.float32)
    guide = _abi.GUIDE_EXECUTED if mode == _abi.MIX_SWITCH else _abi.GUIDE_POLICY
    oracle_lib.orc_mix(n, mode, guide, fptr(f32(step["actions"])), fptr(f32(step["ac_left"])),
                       fptr(f32(step["ac_right"])), fptr(f32(step["gamma"])), fptr(f32(step["rand_vals"])), 3.0, 4.0,
                       fptr(tq), fptr(wt), fptr(gd))
    assert np.array_equal(tq, step[prefix + "_torques"])        # KAMMA.py:632-636 / KA_DDPG_4D.py:627-631
    assert np.array_equal(wt, step[prefix + "_wheel_torqs"])    # KAMMA.py:649
    assert (np.abs(wt) == 4.0).any()                            # the clip is exercised
    np.testing.assert_allclose(gd, step[prefix + "_guiding"], rtol=1e-6, atol=1e-6)  # torch.norm order


def test_is_done_masks_bit_exact(oracle_lib, step):
    n = len(step["reset"])
    rs, to = np.zeros(n, np.uint8), np.zeros(n, np.uint8)
    oracle_lib.orc_is_done(n, iptr(np.ascontiguousarray(step["episode_buf"])), 100, fptr(f32(step["quat"])),
                           fptr(f32(step["base_pos"])), 50.0, u8ptr(rs), u8ptr(to))
    assert np.array_equal(to, step["timeout"])  # KAMMA.py:740-744 (>= 99)
    assert np.array_equal(rs, step["reset"])    # fallen | timeout | out of bounds, KAMMA.py:749-760
    assert 0 < rs.sum() < n and 0 < to.sum() < n


@pytest.mark.parametrize("tag,terminal", [("t0", 0.0), ("t15", 1.5)])
def test_calculate_metrics(oracle_lib, step, tag, terminal):
    n = len(step["reset"])
    prev_lin, prev_ang = f32(step["prev_lin_in"]).copy(), f32(step["prev_ang_in"]).copy()
    sums = f32(step["sums_in"]).copy()
    lin, ang, rew = (np.zeros(n, np.float32) for _ in range(3))
    oracle_lib.orc_metrics(n, fptr(f32(step["v_forward_projected"])), fptr(f32(step["base_velocities"][:, 5])),
                           fptr(prev_lin), fptr(prev_ang), fptr(f32(step["v_delta"])), fptr(f32(step["omega_delta"])),
                           fptr(f32(step["switch_wheel_torqs"])), fptr(f32(step["switch_guiding"])),
                           fptr(f32(step["gamma"])), u8ptr(np.ascontiguousarray(step["reset"])),
                           u8ptr(np.ascontiguousarray(step["timeout"])), np.float32(0.1), -25.0, -0.02, -0.09, 0.2, 0.01,
                           terminal, fptr(lin), fptr(ang), fptr(rew), fptr(sums))
    assert np.array_equal(lin, step[f"lin_acc_{tag}"])  # KAMMA.py:770
    assert np.array_equal(ang, step[f"ang_acc_{tag}"])  # KAMMA.py:771
    assert np.array_equal(prev_lin, f32(step["v_forward_projected"]))  # :774
    # r3 = torch.sum over 4 wheels: order-dependent -> 1e-6; with terminal != 0 the reference's
    # `~timeout_buf` on an int64 tensor contributes terminal * reset * (-1 | -2)  (KAMMA.py:801)
    np.testing.assert_allclose(rew, step[f"rew_{tag}"], rtol=2e-6, atol=2e-6)
    np.testing.assert_allclose(sums, step[f"sums_out_{tag}"], rtol=2e-6, atol=2e-5)
    if terminal:
        r0 = step["rew_t0"]
        delta = rew - r0
        both = (step["reset"] == 1) & (step["timeout"] == 1)
        only = (step["reset"] == 1) & (step["timeout"] == 0)
        np.testing.assert_allclose(delta[both], -2 * terminal, atol=1e-3)
        np.testing.assert_allclose(delta[only], -1 * terminal, atol=1e-3)


def test_forward_projection(oracle_lib, step):
    n = len(step["v_fwd"])
    out = np.zeros(n, np.float32)
    oracle_lib.orc_forward_projection(n, fptr(f32(step["quat"])), fptr(f32(step["base_velocities"][:, :3])), fptr(out))
    np.testing.assert_allclose(out, step["v_fwd"], rtol=1e-6, atol=1e-6)  # KAMMA.py:701-710 (norm / sum order)


@pytest.mark.parametrize("fixture,linear_x", [("refmath_reset.npz", [1.0, 2.0]), ("refmath_reset_small_cmd.npz", [0.0, 0.4])])
def test_reset_draws(oracle_lib, fixture, linear_x):
    from options_in_simulation_amd import default_task_cfg
    d = np.load(os.path.join(G, fixture))
    tc = default_task_cfg()
    tc["task"]["env"]["randomCommandVelocityRanges"]["linear_x"] = linear_x
    cfg = build_config(tc, variant="KAMMA_medium", num_envs=1)
    n = len(d["u"])
    pos, quat, cmd = np.zeros((n, 3), np.float32), np.zeros((n, 4), np.float32), np.zeros((n, 4), np.float32)
    for i in range(n):
        u = np.ascontiguousarray(d["u"][i], np.float32)
        oracle_lib.orc_reset_draws(C.byref(cfg), fptr(u), fptr(pos[i]), fptr(quat[i]), fptr(cmd[i]))
    assert np.array_equal(pos, d["base_pos"])           # KAMMA.py:422-424
    assert np.array_equal(cmd, d["commands"])           # KAMMA.py:448-460, 490-496, 569-575
    assert (cmd[:, 2] == 0).all() and (cmd[:, 0] >= linear_x[0]).all()
    # the yaw quaternion goes through the oracle's deterministic sincos (vs torch.sin/cos): 2e-7
    np.testing.assert_allclose(quat, d["base_quat"], rtol=0, atol=3e-7)  # KAMMA.py:426-434
    if linear_x[1] < 1.0:  # the 'set small commands to zero' branch (KAMMA.py:458) must be covered
        zeroed = (d["commands"][:, 1] == 0) & (d["commands"][:, 3] == 0)
        assert 0 < zeroed.sum() < n


def test_quat_helpers(oracle_lib):
    rng = np.random.default_rng(0)
    for _ in range(50):
        a, b = rng.normal(size=4).astype(np.float32), rng.normal(size=4).astype(np.float32)
        a /= np.linalg.norm(a); b /= np.linalg.norm(b)
        out = np.zeros(4, np.float32)
        oracle_lib.orc_quat_mul(fptr(a), fptr(b), fptr(out))
        # Hamilton product, wxyz
        w1, x1, y1, z1 = a.astype(np.float64); w2, x2, y2, z2 = b.astype(np.float64)
        ref = np.array([w1*w2 - x1*x2 - y1*y2 - z1*z2, w1*x2 + x1*w2 + y1*z2 - z1*y2,
                        w1*y2 - x1*z2 + y1*w2 + z1*x2, w1*z2 + x1*y2 - y1*x2 + z1*w2])
        np.testing.assert_allclose(out, ref, atol=5e-7)
        v = rng.normal(size=3).astype(np.float32)
        o3 = np.zeros(3, np.float32)
        oracle_lib.orc_quat_apply(fptr(a), fptr(v), fptr(o3))
        w, x, y, z = a.astype(np.float64)
        R = np.array([[1-2*(y*y+z*z), 2*(x*y-w*z), 2*(x*z+w*y)], [2*(x*y+w*z), 1-2*(x*x+z*z), 2*(y*z-w*x)],
                      [2*(x*z-w*y), 2*(y*z+w*x), 1-2*(x*x+y*y)]])
        np.testing.assert_allclose(o3, R @ v, atol=1e-6)


def test_extras_means_follow_reference():
    """KAMMA.py:470-488 through the whole oracle env: put known sums into the buffers of a
    one-group env, force a reset of the listed arenas and compare extras with the torch numbers."""
    d = np.load(os.path.join(G, "refmath_extras.npz"))
    from oracle.binding import OracleEnv
    n = len(d["sums"])
    # gamma2 constant 0.375, every arena times out at the next step (ep_buf = 98)
    cfg = build_config(variant="KAMMA_medium", num_envs=n, seed=1, gamma1_mode=_abi.GAMMA_CONST,
                       gamma2_mode=_abi.GAMMA_CONST, gamma1_const=0.375, gamma2_const=0.375)
    env = OracleEnv(cfg)
    env.reset(1)
    # step once with zero action to learn this step's additive terms, then rewind and preload
    ids = d["ids"]
    env.buffer(_abi.BUF_COUNTERS)[:, 0] = 0
    env.buffer(_abi.BUF_COUNTERS)[ids, 0] = 98  # only `ids` time out
    before = env.state()
    env.step(np.zeros((n, 4), np.float32))
    added_a = env.buffer(_abi.BUF_SUMS_A).copy(); added_b = env.buffer(_abi.BUF_SUMS_B).copy()
    # sums of reset arenas were zeroed; recover the per-step addends from a second identical env
    env2 = OracleEnv(cfg); env2.reset(1)
    env2.buffer(_abi.BUF_COUNTERS)[:, 0] = 0
    env2.step(np.zeros((n, 4), np.float32))
    add = np.concatenate([env2.buffer(_abi.BUF_SUMS_A), env2.buffer(_abi.BUF_SUMS_B)], axis=1)
    # third env: preload sums so that after this step's addends they equal the fixture's sums
    env3 = OracleEnv(cfg); env3.reset(1)
    env3.buffer(_abi.BUF_COUNTERS)[:, 0] = 0
    env3.buffer(_abi.BUF_COUNTERS)[ids, 0] = 98
    pre = d["sums"] - add
    env3.buffer(_abi.BUF_SUMS_A)[:] = pre[:, :4]; env3.buffer(_abi.BUF_SUMS_B)[:] = pre[:, 4:]
    env3.step(np.zeros((n, 4), np.float32))
    assert env3.done.sum() == len(ids)
    ei = env3.episode_info()
    assert ei[_abi.EI_COUNT] == len(ids)
    tol = dict(rtol=2e-4, atol=2e-4)  # pre + add reproduces the fixture sums only to ~1 ulp each
    for j in range(8):
        np.testing.assert_allclose(ei[1 + j], d[f"mean_sum{j}"], **tol)
    for slot, key in zip(range(9, 16), ["r1", "r2", "r3", "dense", "sparse", "guiding", "observed"]):
        np.testing.assert_allclose(ei[slot], d[f"mean_{key}_over_final"], rtol=2e-3, atol=2e-3)
    assert ei[16] == np.float32(0.375) and ei[17] == np.float32(0.375) and ei[18] == 0.0
    del before, added_a, added_b


def test_eval_sums_follow_the_fixture(oracle_lib):
    """KAMMA_eval.py:677-685 / :840 on the fixture's sequence: orc_eval_sums step by step, old_torques kept as the
    reference keeps them (the step's clipped torques; zeros for EVERY arena after a step in which any arena reset)."""
    d = np.load(os.path.join(G, "refmath_evalsums.npz"))
    T, n = d["desired_v"].shape
    sums = np.zeros((n, 4), np.float32)
    old = np.zeros((n, 4), np.float32)
    for t in range(T):
        oracle_lib.orc_eval_sums(n, fptr(old), fptr(f32(d["wheel_torqs"][t])), fptr(f32(d["desired_v"][t])),
                                 fptr(f32(d["v_delta"][t])), fptr(sums))
        old = f32(d["wheel_torqs"][t]).copy()
        ids = d["reset"][t].astype(bool)
        if ids.any():
            np.testing.assert_allclose(sums[ids, :3].mean(0) / 10.0, d["rew"][t], rtol=2e-6)  # :482-484
            sums[ids] = 0.0
            old[:] = 0.0  # :470, all arenas
        else:
            assert np.isnan(d["rew"][t]).all()
        assert np.array_equal(old, d["old_torques"][t])
        np.testing.assert_allclose(sums[:, :3], d["sums"][t], rtol=2e-6, atol=1e-6)  # torch.sum over the wheels: order
    assert d["reset"].sum(1).max() == n and ((d["reset"].sum(1) > 0) & (d["reset"].sum(1) < n)).any()


@pytest.mark.parametrize("variant", ["KAMMA_eval", "KA_DDPG_1D_eval"])
def test_eval_sums_through_the_env_with_partial_resets(variant):
    """The whole oracle env against the transcription (tests/golden/make_golden.py evalsums_sequence) on the env's own
    per-step torques, commands and resets: a robot that falls alone zeroes old_torques of every arena (the reference's
    reset_idx, KAMMA_eval.py:470), so the next step's Smoothness of the OTHER arenas is the square of their torques."""
    import torch
    from oracle.binding import OracleEnv
    from tests.golden.make_golden import evalsums_sequence
    n, T = 7, 130
    cfg = build_config(variant=variant, num_envs=n, seed=5)
    env = OracleEnv(cfg)
    env.reset(5)
    rng = np.random.default_rng(0)
    wt, dv, vd, rs, sc, info = [], [], [], [], [], []
    for t in range(T):
        if t in (20, 21, 60):  # arena 3 falls alone (twice in a row), later arena 5
            env.buffer(_abi.BUF_QUAT)[3 if t < 60 else 5] = np.array([0.0, 1.0, 0.0, 0.0], np.float32)
        dv.append(env.buffer(_abi.BUF_COMMANDS)[:, 0].copy())  # desired_v is a view of commands: pre-reset value
        env.step(rng.uniform(-1, 1, (n, cfg.action_dim)).astype(np.float32))
        wt.append(env.buffer(_abi.BUF_WHEEL_TORQS).copy()); vd.append(env.buffer(_abi.BUF_OBS)[:, 2].copy())
        rs.append(env.done.astype(bool).copy()); sc.append(env.buffer(_abi.BUF_SUMS_C)[:, :3].copy())
        info.append(env.episode_info().copy() if env.done.any() else None)
    rs = np.array(rs)
    assert rs[20, 3] and rs[20].sum() == 1 and rs[21].sum() == 1 and rs[60, 5] and rs[60].sum() == 1 and rs[98].sum() == n - 2
    sums, old, rew = evalsums_sequence(torch.from_numpy(np.array(wt)), torch.from_numpy(np.array(dv)),
                                       torch.from_numpy(np.array(vd)), torch.from_numpy(rs))
    np.testing.assert_allclose(np.array(sc), sums.numpy(), rtol=3e-6, atol=1e-6)
    for t in range(T):
        if info[t] is not None:
            np.testing.assert_allclose(info[t][20:23], rew[t].numpy(), rtol=3e-6, atol=1e-7)
    # the global zeroing is visible: at step 21 the arenas that did not reset at step 20 add sum(wt^2) / 10
    others = [i for i in range(n) if i != 3]
    step21 = np.array(sc)[21, others, 0] - np.array(sc)[20, others, 0]
    np.testing.assert_allclose(step21, (np.array(wt)[21, others] ** 2).sum(1) / 10.0, rtol=1e-4)  # a difference of running sums
