"""The Python mirror of the reference's interface (VecEnv / TorqueDistributionTask, options_in_simulation_amd/vecenv.py) under
random READ patterns, against the CPU oracle (GPU box): per sequence a VecEnv of a random size / variant / copy_outputs, STEPS
steps with random actions, and after each step a random choice of what the trainer looks at — nothing (long gaps switch the eager
logging launch off again), `task.observed_components` (26 env-0 values), `infos["episode"]` (every key, `.item()`), both, in either
order, the step's returns. Everything read must be the oracle's value for THAT step, bit for bit; a stale cache would show here.
    python tools/wrapper_fuzz.py [SEQUENCES] [STEPS]"""
import math
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from options_in_simulation_amd import VecEnv, _abi  # noqa: E402
from options_in_simulation_amd.config import VARIANTS  # noqa: E402
from options_in_simulation_amd.vecenv import observed_components_from_diag  # noqa: E402
from oracle.binding import OracleEnv  # noqa: E402

SEQ = int(sys.argv[1]) if len(sys.argv) > 1 else 100
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 120


def eq(a, b):
    return a == b or (isinstance(a, float) and isinstance(b, float) and math.isnan(a) and math.isnan(b))


variants = sorted(VARIANTS)
t0 = time.time()
reads = 0
for seq in range(SEQ):
    rng = np.random.default_rng(5100 + seq)
    N = int(rng.choice([1, 3, 64, 200, 1500, 9000]))
    variant = variants[rng.integers(len(variants))]
    copy_outputs = bool(rng.integers(2))
    env = VecEnv(num_envs=N, device="cuda:0", variant=variant, seed=int(rng.integers(1 << 30)), copy_outputs=copy_outputs,
                 max_episode_length=int(rng.integers(5, 40)))
    cfg = env.task.cfg_native
    orc = OracleEnv(cfg, threads=8)
    seed = int(rng.integers(1 << 30))
    obs, _ = env.reset(seed=seed)
    orc.reset(seed); orc.step(np.zeros((N, cfg.action_dim), np.float32))   # VecEnv.reset: post_reset + one zero-action step
    assert np.array_equal(obs.cpu().numpy(), orc.obs), f"seq {seq}: reset observation"
    eval_keys, single = bool(cfg.eval_sums), cfg.mix_mode == _abi.MIX_BLEND
    layout = _abi.episode_info_layout(eval_keys, single)
    lo = -1.0 if variant.startswith("KA_DDPG") else 0.0
    quiet = 0
    for t in range(STEPS):
        a = rng.uniform(lo, 1.0, size=(N, cfg.action_dim)).astype(np.float32)
        obs, rew, term, trunc, infos = env.step(torch.from_numpy(a).cuda())
        orc.step(a)
        if quiet > 0:           # a stretch without reads (> 16 steps switches the eager logs launch off)
            quiet -= 1
            continue
        r = rng.random()
        if r < 0.08:
            quiet = int(rng.integers(1, 30))
            continue
        order = [x for x in ("diag", "infos", "returns") if rng.random() < 0.6]
        rng.shuffle(order)
        for what in order:
            reads += 1
            where = f"seq {seq} ({variant}, N={N}, copy_outputs={copy_outputs}) step {t}: {what}"
            if what == "diag":
                got, want = env.task.observed_components, observed_components_from_diag(orc.buffer(_abi.BUF_ENV0_DIAG))
                assert list(got) == list(want) and all(eq(got[k], want[k]) for k in want), where
            elif what == "infos":
                ei = orc.episode_info()
                ep = infos["episode"]
                assert list(ep.keys()) == [k for k, _ in layout], where
                for k, slot in layout:
                    assert eq(float(ep[k].item()), float(ei[slot])), f"{where} [{k}]: {float(ep[k].item())} vs {float(ei[slot])}"
            else:
                torch.cuda.synchronize()
                assert np.array_equal(obs.cpu().numpy(), orc.obs) and np.array_equal(rew.cpu().numpy().reshape(-1), orc.rew.reshape(-1)), where
                assert np.array_equal(term.cpu().numpy().reshape(-1).astype(np.uint8), orc.done.reshape(-1)) and not bool(trunc.any()), where
    env.close(); orc.close()
    if seq % 10 == 9:
        print(f"ok {seq + 1} sequences, {reads} reads", flush=True)
print(f"{SEQ} sequences x {STEPS} steps, {reads} reads in random patterns, all equal to the oracle, {time.time() - t0:.0f} s")
