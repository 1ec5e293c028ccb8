"""The logging path over many steps (GPU box): an env-only loop that reads task.observed_components and every value of
infos["episode"] after EVERY step — the published block is polled 26 500 times at three sizes (one workgroup / one wave per chunk).
Prints the wall time per logged step and how many reset events the record went through."""
import sys, os, time, math
sys.path.insert(0, os.getcwd())
import torch
from options_in_simulation_amd import VecEnv
for N, steps in ((4096, 20000), (65536, 5000), (140000, 1500)):
    env = VecEnv(num_envs=N, device="cuda:0", variant="KAMMA_medium", seed=3)
    states, _ = env.reset()
    a = torch.rand((N, 4), device="cuda")
    t0 = time.perf_counter(); last_count = None; events = 0
    for t in range(steps):
        states, rew, term, trunc, infos = env.step(a)
        oc = env.task.observed_components
        ep = {k: v.item() for k, v in infos["episode"].items()}
        assert math.isfinite(oc["env0_current_v"]) and oc["env0_episode_count"] >= 1
        c = env.task.native.host_logs()[0][19]   # OIS_EI_STEP: the step of the newest reset event
        if c != last_count: events += 1; last_count = c
    torch.cuda.synchronize()
    print(f"N={N}: {steps} logged steps, {(time.perf_counter()-t0)/steps*1e6:.1f} us/step, {events} changes of the reset-event record, episode_count {oc['env0_episode_count']}", flush=True)
    env.close()
