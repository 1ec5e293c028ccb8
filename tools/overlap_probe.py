"""Does splitting the 65 536 arenas into two shards on two streams (actor of one shard overlapping
the env step of the other) beat the single-stream loop? (GPU box)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from options_in_simulation_amd import _abi, build_config
from options_in_simulation_amd.vecenv import NativeEnv

N = 65536
torch.manual_seed(0)
actor = torch.nn.Sequential(torch.nn.Linear(6, 512), torch.nn.ELU(), torch.nn.Linear(512, 512), torch.nn.ELU(),
                            torch.nn.Linear(512, 128), torch.nn.ELU(), torch.nn.Linear(128, 4), torch.nn.Sigmoid()).cuda()


def single(steps):
    env = NativeEnv(build_config(variant="KAMMA_medium", num_envs=N, seed=42), "cuda:0")
    env.reset(42)
    obs = env.buffer(_abi.BUF_OBS)
    for _ in range(20):
        with torch.no_grad():
            env.step(actor(obs))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps):
        with torch.no_grad():
            env.step(actor(obs))
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e6


def split(steps, parts):
    n = N // parts
    envs = [NativeEnv(build_config(variant="KAMMA_medium", num_envs=n, seed=42, arena_offset=i * n), "cuda:0") for i in range(parts)]
    streams = [torch.cuda.Stream() for _ in range(parts)]
    for e in envs:
        e.reset(42)
    torch.cuda.synchronize()

    def run(k):
        for _ in range(k):
            for e, s in zip(envs, streams):
                with torch.cuda.stream(s), torch.no_grad():
                    e.step(actor(e.buffer(_abi.BUF_OBS)))
    run(20)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    run(steps)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e6


print(f"single stream: {single(300):.1f} us per 65 536-arena step", flush=True)
for p in (2, 4):
    print(f"{p} shards on {p} streams: {split(300, p):.1f} us per 65 536-arena step", flush=True)
print(f"single stream again: {single(300):.1f} us", flush=True)
