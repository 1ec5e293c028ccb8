import os, sys, torch
sys.path.insert(0, os.getcwd())
from options_in_simulation_amd import _abi, build_config
from options_in_simulation_amd.vecenv import NativeEnv
N=65536
env = NativeEnv(build_config(variant="KA_DDPG_4D", num_envs=N, seed=1), "cuda:0"); env.reset(1)
a = torch.rand((N,4), device="cuda")*2-1
def t(n=300):
    for _ in range(20): env.step(a)
    torch.cuda.synchronize(); e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): env.step(a)
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)*1e3/n
plain=t(); env.attach_replay(64); with_rp=t()
print(f"ois_step at 65536 arenas: {plain:.1f} us plain, {with_rp:.1f} us with the transition memory attached (+72 B per arena-step)")
