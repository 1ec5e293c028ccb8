"""What one step of the reference's own statements (tests/golden/refexec_cosim.npz, made by
tests/golden/make_golden_cosim.py) must equal on the oracle's side. Shared by the generator — which checks while it
records — and by tests/test_refexec_cosim.py, which replays the stored inputs through the oracle.

Exact: masks, counters, commands, clipped torques, the observation columns (element-wise float32 chains on the same
inputs). 3e-6: quantities behind a reduction over the 4 wheels or the 3 axes, whose order torch does not define
(guiding reward, r3, the projection). 2e-5: running sums of up to 99 such terms and means over the resetting arenas."""
import numpy as np

from options_in_simulation_amd import _abi
from options_in_simulation_amd.vecenv import observed_components_from_diag


def check_step(orc, cfg, out, tag, first=False, drawn_commands=False):
    """drawn_commands: the commands come out of torch.normal(mean, std) every step (the per-level curriculum), which is
    `mean + std z` evaluated in double inside torch and in float32 here: an ulp or two apart, and with them everything the
    controller derives from the command"""
    A = _abi
    if drawn_commands:
        eq = lambda a, b, what: np.testing.assert_allclose(a, b, rtol=3e-6, atol=3e-6, err_msg=f"{tag}: {what}")  # noqa: E731
    else:
        eq = lambda a, b, what: np.testing.assert_array_equal(a, b, err_msg=f"{tag}: {what}")  # noqa: E731
    exact = lambda a, b, what: np.testing.assert_array_equal(a, b, err_msg=f"{tag}: {what}")  # noqa: E731

    def close(a, b, what, rtol=3e-6, atol=3e-6):
        np.testing.assert_allclose(a, b, rtol=rtol, atol=atol, equal_nan=True, err_msg=f"{tag}: {what}")
    cnt = orc.buffer(A.BUF_COUNTERS)
    exact(cnt[:, 0], out["episode_buf"], "episode_buf (KAMMA.py:462, 675)")
    exact(cnt[:, 1], out["sim_steps"], "sim_steps (:676)")
    exact(cnt[:, 2], out["episode_count"], "episode_count (:463)")
    eq(orc.buffer(A.BUF_COMMANDS), out["commands"], "commands (:448-460, 490-496)")
    if not first:
        exact(orc.done, out["reset"], "reset_buf (:736-760)")
        exact(orc.timeout, out["timeout"], "timeout_buf (:740-744)")
        if "fraction" in out and drawn_commands:
            close(orc.buffer(A.BUF_CMD_FRACTION), out["fraction"], "fraction (:498)", rtol=3e-4, atol=3e-5)
        eq(orc.buffer(A.BUF_WHEEL_TORQS), out["wheel_torqs"], "wheel_torqs (:632-636, 649)")
        eq(orc.obs, np.clip(out["obs"], -cfg.clip_obs, cfg.clip_obs), "obs_buf (:833; rl_task.py:99)")
        # (drawn commands: v_delta = cmd - v carries the command's ulp as a RELATIVE error of ulp / |v_delta|, squared and x 25 in r1)
        loose = dict(rtol=3e-4, atol=3e-5) if drawn_commands else {}
        close(orc.rew, out["rew"], "rew_buf (:767-801)", **loose)
        sums = np.concatenate([orc.buffer(A.BUF_SUMS_A), orc.buffer(A.BUF_SUMS_B)], 1)
        close(sums, out["sums"], "episode_sums (:803-810, 485)", **(loose or dict(rtol=2e-5, atol=2e-5)))
        eq(orc.buffer(A.BUF_LINVEL)[:, 3], out["prev_lin"], "previous_linear_velocity (:774)")
        eq(orc.buffer(A.BUF_ANGVEL)[:, 3], out["prev_ang"], "previous_angular_velocity (:775)")
        close(orc.buffer(A.BUF_POSE)[:, 3], out["v_fwd"], "v_forward_projected (:704-710)", rtol=2e-6, atol=2e-7)
        if "sums_eval" in out:
            close(orc.buffer(A.BUF_SUMS_C)[:, :3], out["sums_eval"], "Smoothness / Tracking error / Desired velocity (KAMMA_eval.py:677-685, 840)",
                  rtol=2e-5, atol=2e-6)
        if "components" in out:   # the 26 env-0 diagnostics (:840-867)
            mine = observed_components_from_diag(orc.buffer(A.BUF_ENV0_DIAG))
            keys = [str(k) for k in out["components_keys"]]
            assert sorted(keys) == sorted(mine), (tag, sorted(set(keys) ^ set(mine)))
            close(np.array([mine[k] for k in keys], np.float64), out["components"], "observed_components (:840-867)",
                  rtol=2e-3 if drawn_commands else 2e-4, atol=2e-4 if drawn_commands else 2e-5)
    if first or out["reset"].any():   # extras["episode"] (:479-488) is rewritten by every reset_idx call
        vec = orc.episode_info()
        mine = dict(A.episode_info_items(vec, bool(cfg.eval_sums), cfg.mix_mode == A.MIX_BLEND))
        keys = [str(k) for k in out["extras_keys"]]
        assert sorted(keys) == sorted(mine), (tag, sorted(set(keys) ^ set(mine)))
        close(np.array([mine[k] for k in keys], np.float64), out["extras"], "extras['episode'] (:470-488)",
              rtol=3e-4 if drawn_commands else 3e-5, atol=3e-4 if drawn_commands else 3e-5)
