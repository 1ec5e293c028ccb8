import os, sys, time, torch
sys.path.insert(0, os.getcwd())
from options_in_simulation_amd import DeviceFIFOMemory, VecEnv
env = VecEnv(num_envs=65536, device="cuda:0", variant="KA_DDPG_4D", seed=1)
s,_ = env.reset()
mem = DeviceFIFOMemory(env, 16)
for _ in range(16): env.step(torch.zeros((65536,4),device="cuda"))
def t(fn,n=100):
    fn(); torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter()-t0)/n*1e6
names=["states","actions","rewards","next_states","terminated"]
print("len(memory)", t(lambda: len(mem)))
print("randperm(1M)[:4096]", t(lambda: torch.randperm(1<<20, device="cuda")[:4096]))
print("randint", t(lambda: torch.randint(0, 1<<20, (4096,), device="cuda")))
print("memory.sample (no replacement)", t(lambda: mem.sample(names, 4096)))
mem.replacement=True
print("memory.sample (replacement)", t(lambda: mem.sample(names, 4096)))
print("observed_components", t(lambda: (env.step(torch.zeros((65536,4),device='cuda')), env.task.observed_components)) , "(incl. step)")
