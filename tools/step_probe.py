"""Single-step launch timing under different host-side conditions (diagnostic)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from options_in_simulation_amd import _abi, build_config
from options_in_simulation_amd.vecenv import NativeEnv
N = 65536
env = NativeEnv(build_config(variant="KAMMA_medium", num_envs=N, seed=42), "cuda:0")
env.reset(42)
a = torch.rand((N, 4), device="cuda")
def timeit(name, fn, n=200):
    for _ in range(20): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); print(f"{name}: {(time.perf_counter() - t) / n * 1e6:.1f} us/step")
timeit("rollout k=1 RANDOM", lambda: env.rollout(1, _abi.POLICY_RANDOM))
timeit("rollout k=1 ZERO", lambda: env.rollout(1, _abi.POLICY_ZERO))
timeit("step(a) external", lambda: env.step(a))
def with_events():
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); env.step(a); e1.record()
timeit("step(a) + 2 events", with_events)
z = torch.zeros((N, 4), device="cuda")
timeit("step(zeros) external", lambda: env.step(z))
half = torch.full((N, 4), 0.5, device="cuda")
timeit("step(0.5) external", lambda: env.step(half))
