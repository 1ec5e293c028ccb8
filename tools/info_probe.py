"""cost of the logging-side kernels at 65 536 arenas: ois_episode_info, ois_stats_fetch (GPU box)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from options_in_simulation_amd import _abi, build_config
from options_in_simulation_amd.vecenv import NativeEnv

N = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
env = NativeEnv(build_config(variant="KAMMA_medium", num_envs=N, seed=42), "cuda:0")
env.reset(42)
env.rollout(100, _abi.POLICY_RANDOM)
torch.cuda.synchronize()


def timeit(name, fn, n=50):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    print(f"N={N} {name}: {e0.elapsed_time(e1) * 1e3 / n:.1f} us per call", flush=True)


timeit("ois_episode_info", lambda: env.episode_info())
timeit("ois_stats_fetch", lambda: env.stats())
timeit("ois_env0_diag (blocking copy)", lambda: env.env0_diag(direct=True))


def logs():
    env.epoch += 1  # a new state as far as the host cache is concerned: every call fetches
    env.host_logs()


timeit("ois_logs_fetch (finalise + publish + wait, nothing else in flight)", logs)
