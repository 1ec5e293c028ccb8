"""The statistic the physics calibration is fitted on and tested with (CALIBRATION, NOT PARITY).

The reference ships two artefacts of runs with PhysX in the loop (my_runs/):
the env-0 window means of a training run, whose first 100 k steps are dominated by the expert
("criteria") controller (gamma 1 -> 0.8), and the README figure's tracking-error-vs-speed curve of
that controller. tools/calibrate_physics.py distils them into tests/golden/ref_stats.json; this
module computes the SAME quantities from an expert-only run (KAMMA_g11_g21: gamma = 1 forever)
of any environment object — the CPU oracle or the HIP path:

* per arena, the mean over a window of 800 consecutive policy steps (the log's write interval,
  train_kamma_4d_nofifo_random_RLIL.py:176) of desired_v, current_v, v_delta (the log's columns,
  KAMMA.py:840-846); their grand means; the least-squares line of window-mean v_delta against
  window-mean desired_v (the log gives 0.309 x - 0.302 over rows with step < 100 000);
* per finished episode, the mean |v_delta| ("rew_Tracking error", KAMMA_eval.py:840) binned by
  the episode's commanded speed (the figure's axes).
"""
import numpy as np

from options_in_simulation_amd import _abi

WINDOW = 800
TE_BINS = (1.0, 1.2, 1.4, 1.6, 1.8, 2.0)


def expert_run_stats(env, n, steps=WINDOW, to_numpy=np.asarray):
    """env: reset already done; has step(np.ndarray[N, A]) and buffer(id). Zero actions: with
    gamma = 1 the switch always executes the expert's torques (KAMMA.py:632-635)."""
    z = np.zeros((n, 4), np.float32)
    vdes, vd, v = np.zeros(n), np.zeros(n), np.zeros(n)
    te_sum, cmd_prev = np.zeros(n), None
    ep_cmd, ep_te = [], []
    for _ in range(steps):
        env.step(z)
        o = to_numpy(env.buffer(_abi.BUF_OBS))
        # obs[:, 0] shows the post-reset command on a reset step: the log's desired_v does too (KAMMA.py:840)
        vdes += o[:, 0]; vd += o[:, 2]; v += o[:, 0] - o[:, 2]
        te_sum += np.abs(o[:, 2])
        d = to_numpy(env.buffer(_abi.BUF_DONE)).astype(bool)
        if d.any() and cmd_prev is not None:
            ep_cmd.append(cmd_prev[d].copy()); ep_te.append(te_sum[d] / 99.0)  # 99-step episodes (KAMMA.py:741)
            te_sum[d] = 0.0
        cmd_prev = o[:, 0].copy()
    vdes /= steps; vd /= steps; v /= steps
    slope, icpt = np.polyfit(vdes, vd, 1)
    cmd, te = np.concatenate(ep_cmd), np.concatenate(ep_te)
    bins = [float(te[(cmd >= lo) & (cmd < hi)].mean()) for lo, hi in zip(TE_BINS[:-1], TE_BINS[1:])]
    return {"desired_v": float(vdes.mean()), "current_v": float(v.mean()), "v_delta": float(vd.mean()),
            "slope": float(slope), "intercept": float(icpt), "te_by_command": bins}
