"""Host-side enqueue cost of ois_step / ois_rollout through the Python binding (diagnostic)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from options_in_simulation_amd import _abi, build_config
from options_in_simulation_amd.vecenv import NativeEnv
for N in (64, 65536):
    env = NativeEnv(build_config(variant="KAMMA_medium", num_envs=N, seed=42), "cuda:0")
    env.reset(42)
    a = torch.rand((N, 4), device="cuda")
    for fn, name in ((lambda: env.rollout(1, _abi.POLICY_RANDOM), "rollout(1)"), (lambda: env.step(a), "step(a)")):
        for _ in range(50): fn()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(2000): fn()
        t_enq = time.perf_counter() - t
        torch.cuda.synchronize()
        t_all = time.perf_counter() - t
        print(f"N={N} {name}: host enqueue {t_enq / 2000 * 1e6:.1f} us/call, end-to-end {t_all / 2000 * 1e6:.1f} us/call")
