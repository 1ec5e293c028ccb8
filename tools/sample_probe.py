"""fused mini-batch gather (ois_replay_sample) vs five torch indexed copies (GPU box)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from options_in_simulation_amd import _abi, build_config
from options_in_simulation_amd.vecenv import NativeEnv

N, M = 65536, 16
env = NativeEnv(build_config(variant="KA_DDPG_4D", num_envs=N, seed=2), "cuda:0")
env.reset(2)
t = env.attach_replay(M)
env.rollout(M, _abi.POLICY_RANDOM)
flat = {k: v.reshape(-1, v.shape[-1]) for k, v in t.items()}


def timeit(fn, n=200):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


for batch in (4096, 65536, 1048576):
    idx = torch.randint(0, M * N, (batch,), device="cuda")
    a = timeit(lambda: env.replay_sample(idx))
    b = timeit(lambda: [flat[k][idx] for k in flat])
    print(f"batch {batch}: fused {a:.1f} us ({69 * 2 * batch / a / 1e3:.0f} GB/s), torch x5 {b:.1f} us", flush=True)
