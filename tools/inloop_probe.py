"""Kernel-only time (ois_profile_*) of single ois_step launches at 65 536 arenas: back to back (state
warm in L2 / MALL) and with a cache-sweeping copy between the steps (what the actor's GEMMs do to the
caches), for several substep counts: the intercept is what a launch costs besides the substeps.
  OIS_LIB_OVERRIDE=<variant .so> python tools/inloop_probe.py      (e.g. a -DOIS_BLOCK=256 build)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from options_in_simulation_amd import build_config
from options_in_simulation_amd.vecenv import NativeEnv
N = int(os.environ.get("PROBE_N", "65536"))
a = torch.rand((N, 4), device="cuda")
src = torch.empty(96 << 20, dtype=torch.float32, device="cuda")  # 384 MiB: more than the 256 MiB MALL
dst = torch.empty_like(src)
for label, sweep in (("back to back", False), ("cache sweep between steps", True)):
    res = {}
    for nsub in (1, 5, 10, 20):
        env = NativeEnv(build_config(variant="KAMMA_medium", num_envs=N, seed=42, decimation=nsub, extra_substeps=0), "cuda:0")
        env.reset(42)
        for _ in range(10):
            env.step(a)
        torch.cuda.synchronize()
        env.profile_begin()
        for _ in range(60):
            if sweep:
                dst.copy_(src)
            env.step(a)
        ms, n = env.profile_end()
        res[nsub] = ms / n * 1e3
        env.close()
    b = (res[20] - res[1]) / 19.0
    print(f"{label:28s} " + " ".join(f"nsub={k}:{v:.2f}" for k, v in res.items()) + f"  -> {res[1] - b:.2f} us + {b:.3f} us/substep", flush=True)
