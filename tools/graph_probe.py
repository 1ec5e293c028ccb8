"""actor + ois_step captured in a HIP graph (torch.cuda.CUDAGraph) vs eager launches (GPU box)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from options_in_simulation_amd import _abi, build_config
from options_in_simulation_amd.vecenv import NativeEnv

torch.manual_seed(0)
actor = torch.nn.Sequential(torch.nn.Linear(6, 512), torch.nn.ELU(), torch.nn.Linear(512, 512), torch.nn.ELU(),
                            torch.nn.Linear(512, 128), torch.nn.ELU(), torch.nn.Linear(128, 4), torch.nn.Sigmoid()).cuda()
for N in (1024, 4096, 16384, 65536):
    env = NativeEnv(build_config(variant="KAMMA_medium", num_envs=N, seed=42), "cuda:0")
    env.reset(42)
    obs = env.buffer(_abi.BUF_OBS)

    def eager(k):
        for _ in range(k):
            with torch.no_grad():
                env.step(actor(obs))
    eager(20); torch.cuda.synchronize(); t0 = time.perf_counter(); eager(300); torch.cuda.synchronize()
    te = (time.perf_counter() - t0) / 300 * 1e6
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s), torch.no_grad():
        for _ in range(3):
            env.step(actor(obs))
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g), torch.no_grad():
        env.step(actor(obs))
    for _ in range(20):
        g.replay()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(300):
        g.replay()
    torch.cuda.synchronize()
    tg = (time.perf_counter() - t0) / 300 * 1e6
    print(f"N={N}: eager {te:.1f} us/step, graph {tg:.1f} us/step", flush=True)
    env.close()
