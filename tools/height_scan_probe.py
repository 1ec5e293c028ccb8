"""ois_height_scan alone (profiling target): 65 536 arenas on the 6 x 2-tile field, random poses after 60 steps"""
import os, sys, json
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from options_in_simulation_amd import _abi, build_config
from options_in_simulation_amd import terrain as tg
from options_in_simulation_amd.vecenv import NativeEnv
N = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
T = tg.Terrain({"numLevels": 6, "numTerrains": 2, "mapLength": 10.0, "mapWidth": 10.0}, N, seed=0)
env = NativeEnv(build_config(variant="KAMMA_eval", num_envs=N, seed=1), "cuda:0")
lt = T.sample_levels_types(N, rng=np.random.RandomState(0))
if "--one-tile" in sys.argv:  # every robot on the same tile: the field the scans touch fits each L2
    lt[:] = 0
env.attach_terrain(T, lt)
env.reset(1)
env.rollout(60, _abi.POLICY_RANDOM)
scan = torch.empty((N, 140), device="cuda")
env.height_scan(scan); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    env.height_scan(scan)
e1.record(); torch.cuda.synchronize()
out = {"arenas": N, "height_scan_us": e0.elapsed_time(e1) * 1e3 / reps, "checksum": float(scan.double().sum())}
# how much of it is line traffic: the same scan with every robot axis-aligned (a row of the row-major field
# holds a whole line of scan points) and with every robot at 45 degrees (every point on another row)
if "--yaw" in sys.argv:
    q = env.buffer(_abi.BUF_QUAT)
    q0 = q.clone()
    def timeit():
        env.height_scan(scan); torch.cuda.synchronize()
        e0.record()
        for _ in range(reps):
            env.height_scan(scan)
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / reps
    out["again_us"] = timeit()
    q[:, 1:3] = 0; q.div_(q.norm(dim=1, keepdim=True))
    out["own_yaw_only_us"] = timeit()
    q.copy_(q0); out["restored_us"] = timeit()
    for name, yaw in (("aligned_yaw0", 0.0), ("yaw90", np.pi / 2), ("yaw45", np.pi / 4)):
        q.zero_(); q[:, 0] = float(np.cos(yaw / 2)); q[:, 3] = float(np.sin(yaw / 2))
        env.height_scan(scan); torch.cuda.synchronize()
        e0.record()
        for _ in range(reps):
            env.height_scan(scan)
        e1.record(); torch.cuda.synchronize()
        out[name + "_us"] = e0.elapsed_time(e1) * 1e3 / reps
print(json.dumps(out))
