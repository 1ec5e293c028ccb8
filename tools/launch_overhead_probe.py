"""time(k) of fused rollouts for several k and N: separates the per-launch cost from the per-step cost"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from options_in_simulation_amd import _abi, build_config
from options_in_simulation_amd.vecenv import NativeEnv

def run(N, ks=(1, 2, 3, 4, 8, 16), reps=100):
    env = NativeEnv(build_config(variant="KAMMA_medium", num_envs=N, seed=42), "cuda:0")
    env.reset(42)
    out = {}
    for k in ks:
        for _ in range(10): env.rollout(k, _abi.POLICY_RANDOM)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): env.rollout(k, _abi.POLICY_RANDOM)
        e1.record(); torch.cuda.synchronize()
        out[k] = e0.elapsed_time(e1) * 1e3 / reps
    ks = sorted(out)
    b = (out[ks[-1]] - out[ks[0]]) / (ks[-1] - ks[0])
    a = out[ks[0]] - b * ks[0]
    print(f"N={N}: " + " ".join(f"k={k}:{out[k]:.1f}" for k in ks) + f"  -> per-launch {a:.1f} us + {b:.1f} us/step", flush=True)
    env.close()

for N in [int(x) for x in (sys.argv[1:] or ['4096', '16384', '32768', '49152', '65536', '131072'])]:
    run(N)
