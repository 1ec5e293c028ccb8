"""Kernel-duration probe for rocprofv3 --kernel-trace: step launches back to back, then the
same launches interleaved with a large GEMM (as in bench.py's actor loop)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from options_in_simulation_amd import _abi, build_config
from options_in_simulation_amd.vecenv import NativeEnv
N = 65536
env = NativeEnv(build_config(variant="KAMMA_medium", num_envs=N, seed=42), "cuda:0")
env.reset(42)
a = torch.rand((N, 4), device="cuda")
x = torch.rand((N, 512), device="cuda"); w = torch.rand((512, 512), device="cuda")
for _ in range(30): env.step(a)
torch.cuda.synchronize()
for _ in range(100): env.step(a)          # phase A: back to back
torch.cuda.synchronize()
for _ in range(100):                      # phase B: a GEMM between steps
    y = x @ w
    env.step(a)
torch.cuda.synchronize()
