"""what the per-level command curriculum adds to a policy step (the extra level_command_kernel launch): wall time of
back-to-back ois_step launches at 65 536 arenas with and without cmd_sampler = OIS_CMD_LEVEL"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from options_in_simulation_amd import _abi, build_config
from options_in_simulation_amd.vecenv import NativeEnv
N = 65536
a = torch.rand((N, 4), device="cuda")
for name, kw in (("RAND sampler", {}), ("LEVEL sampler", {"cmd_sampler": _abi.CMD_LEVEL})):
    env = NativeEnv(build_config(variant="KAMMA_medium", num_envs=N, seed=42, **kw), "cuda:0")
    env.reset(42)
    for _ in range(20):
        env.step(a)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(300):
        env.step(a)
    torch.cuda.synchronize()
    print(f"{name}: {(time.perf_counter() - t0) / 300 * 1e6:.2f} us per policy step (wall, back to back ois_step)", flush=True)
    # 20-step rollouts: under the level sampler k single-step launches + ONE command launch (the continuation launches
    # re-draw the commands of the step before them in their prologue)
    for _ in range(3):
        env.rollout(20, _abi.POLICY_RANDOM)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(30):
        env.rollout(20, _abi.POLICY_RANDOM)
    torch.cuda.synchronize()
    print(f"{name}: {(time.perf_counter() - t0) / 600 * 1e6:.2f} us per policy step in 20-step rollouts", flush=True)
    env.close()
