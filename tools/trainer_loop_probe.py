"""Per-step wall time of the loop the reference trainer runs around the env
(src/my_trainers/base_KA_DDPG.py:170-262): actor -> env.step -> 26 env-0 diagnostics ->
executed actions -> every value of infos["episode"] read with .item()."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from options_in_simulation_amd import VecEnv
from options_in_simulation_amd.tracking import ENV0_TAGS

N = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
env = VecEnv(num_envs=N, device="cuda:0", variant="KAMMA_medium", seed=1)
torch.manual_seed(0)
actor = torch.nn.Sequential(torch.nn.Linear(6, 512), torch.nn.ELU(), torch.nn.Linear(512, 512), torch.nn.ELU(),
                            torch.nn.Linear(512, 128), torch.nn.ELU(), torch.nn.Linear(128, 4), torch.nn.Sigmoid()).cuda()


def loop(steps, read_diag, read_infos, on_host=True):
    env.task.infos_on_host = on_host
    states, _ = env.reset()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps):
        with torch.no_grad():
            actions = actor(states)
        states, rew, term, trunc, infos = env.step(actions)
        if read_diag:
            oc = env.task.observed_components
            _ = [oc[k] for k in ENV0_TAGS]
            _ = env.task.wheel_torqs / env.task.action_scale
        if read_infos and "episode" in infos:
            for k, v in infos["episode"].items():
                if isinstance(v, torch.Tensor) and v.numel() == 1:
                    v.item()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e6


loop(20, True, True)
print(f"N={N} actor + step only:                      {loop(200, False, False):8.1f} us/step")
print(f"N={N} + observed_components + executed actions: {loop(200, True, False):8.1f} us/step")
print(f"N={N} + infos['episode'] .item() x22 (host copy): {loop(200, True, True):8.1f} us/step")
print(f"N={N} + infos['episode'] .item() x22 (device tensors, as the reference returns them): {loop(200, True, True, on_host=False):8.1f} us/step")
