"""cost of the height-field contact: fused-rollout us/step on the plane vs on a terrain, with the wheel footprints sampled at every
substep and (ois_terrain.footprint_stride = 2) at every second one; the height scan (GPU box)"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from options_in_simulation_amd import _abi, build_config  # noqa: E402
from options_in_simulation_amd import terrain as tg  # noqa: E402
from options_in_simulation_amd.vecenv import NativeEnv  # noqa: E402


def timed(env, k=20, reps=10):
    env.rollout(k, _abi.POLICY_RANDOM)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        env.rollout(k, _abi.POLICY_RANDOM)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (reps * k)


out = {}
N = 65536
T = tg.Terrain({"numLevels": 6, "numTerrains": 2, "mapLength": 10.0, "mapWidth": 10.0}, N, seed=0)
lt = T.sample_levels_types(N, rng=np.random.RandomState(0))
for packed in ("1", "0"):
    os.environ["OIS_PACKED"] = packed
    cfg = build_config(variant="KAMMA_eval", num_envs=N, seed=1, eval_sums=0)  # gamma = 0 (random torques executed); without the eval sums, whose rollouts are single-step launches
    env = NativeEnv(cfg, "cuda:0")
    env.reset(1)
    flat = timed(env)
    env.attach_terrain(T, lt)
    env.reset(1)
    ter = timed(env)
    env.attach_terrain(T, lt, footprint_stride=2)
    env.reset(1)
    ter2 = timed(env)
    env.attach_terrain(T, lt)
    env.reset(1)
    scan = torch.empty((N, 140), device="cuda")
    env.height_scan(scan); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        env.height_scan(scan)
    e1.record(); torch.cuda.synchronize()
    out["packed" if packed == "1" else "scalar"] = {"flat_us_per_step": round(flat, 2), "terrain_us_per_step": round(ter, 2), "terrain_footprint_stride_2_us_per_step": round(ter2, 2),
                                                      "height_scan_us": round(e0.elapsed_time(e1) * 1e3 / 20, 2)}
    print(json.dumps(out), flush=True)
out["field_MB"] = round(T.heightsamples.nbytes / 1e6, 1)
json.dump(out, open("gpurun_out/terrain_probe.json", "w"), indent=1)
