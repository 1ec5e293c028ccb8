"""Launch a few fused k-step rollouts (one long dispatch each) — a profiling target whose
dispatches are long enough (>1 ms) for rocprofv3 PMC counters and clock estimates."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from options_in_simulation_amd import _abi, build_config  # noqa: E402
from options_in_simulation_amd.vecenv import NativeEnv  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--arenas", type=int, default=65536)
ap.add_argument("--k", type=int, default=20)
ap.add_argument("--launches", type=int, default=5)
ap.add_argument("--variant", default="KAMMA_medium")
ap.add_argument("--terrain", type=int, default=0, help="1 | 2: on the height field of bench.py --terrain, with that footprint stride")
a = ap.parse_args()
env = NativeEnv(build_config(variant=a.variant, num_envs=a.arenas, seed=42), "cuda:0")
if a.terrain:
    import numpy as np
    from options_in_simulation_amd import terrain as tg
    T = tg.Terrain({"numLevels": 6, "numTerrains": 2, "mapLength": 10.0, "mapWidth": 10.0}, a.arenas, seed=0)
    env.attach_terrain(T, T.sample_levels_types(a.arenas, rng=np.random.RandomState(0)), footprint_stride=a.terrain)
env.reset(42)
env.rollout(a.k, _abi.POLICY_RANDOM)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.launches):
    env.rollout(a.k, _abi.POLICY_RANDOM)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"arenas={a.arenas} k={a.k}: {dt / a.launches / a.k * 1e6:.2f} us per policy step, "
      f"{a.arenas * a.k * a.launches / dt / 1e6:.1f} M env-steps/s")
