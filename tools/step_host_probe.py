import sys, time, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from options_in_simulation_amd import VecEnv, default_task_cfg
for N in (1, 4096):
    env = VecEnv(default_task_cfg(), num_envs=N, device="cuda:0", variant="KAMMA_medium", seed=1)
    obs, _ = env.reset()
    a = torch.rand((N, 4), device="cuda")
    for _ in range(200): env.step(a)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 3000
    for _ in range(n): obs, r, term, trunc, info = env.step(a)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"N={N}: {dt / n * 1e6:.1f} us per env.step() wall ({n / dt:.0f} steps/s)")
    # raw native step
    nat = env.task.native
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): nat.step(a)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"N={N}: {dt / n * 1e6:.1f} us per NativeEnv.step() wall")
    env.close()
