"""Per-step wall time of the loop the reference trainer runs around the env
(src/my_trainers/base_KA_DDPG.py:170-262): actor -> env.step -> 26 env-0 diagnostics ->
executed actions -> every value of infos["episode"] read with .item().
    python tools/trainer_loop_probe.py [N] [--breakdown]     --breakdown: where the host spends a step of the full loop"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from options_in_simulation_amd import VecEnv
from options_in_simulation_amd.tracking import ENV0_TAGS

args = [a for a in sys.argv[1:] if not a.startswith("--")]
N = int(args[0]) if args else 65536
env = VecEnv(num_envs=N, device="cuda:0", variant="KAMMA_medium", seed=1)
torch.manual_seed(0)
actor = torch.nn.Sequential(torch.nn.Linear(6, 512), torch.nn.ELU(), torch.nn.Linear(512, 512), torch.nn.ELU(),
                            torch.nn.Linear(512, 128), torch.nn.ELU(), torch.nn.Linear(128, 4), torch.nn.Sigmoid()).cuda()
clock = time.perf_counter
STEPS = 500  # per line; a line moves by a few us from run to run and by up to 15 us from box to box (the synchronisation's wake-up)


def loop(steps, read_diag, read_infos, on_host=True, phases=None, bare_sync=False):
    env.task.infos_on_host = on_host
    states, _ = env.reset()
    torch.cuda.synchronize(); t0 = clock()
    for _ in range(steps):
        ta = clock()
        with torch.no_grad():
            actions = actor(states)
        tb = clock()
        states, rew, term, trunc, infos = env.step(actions)
        tc = clock()
        if bare_sync:
            torch.cuda.current_stream().synchronize()
        if read_diag:
            oc = env.task.observed_components  # base_KA_DDPG.py:184-209: 26 lookups of the same property
            _ = [env.task.observed_components[k] for k in ENV0_TAGS]
            td = clock()
            _ = env.task.wheel_torqs / env.task.action_scale  # :212
        else:
            td = clock()
        te = clock()
        if read_infos and "episode" in infos:
            for k, v in infos["episode"].items():  # :259-262
                if isinstance(v, torch.Tensor) and v.numel() == 1:
                    v.item()
        tf = clock()
        if phases is not None:
            for i, d in enumerate((tb - ta, tc - tb, td - tc, te - td, tf - te)):
                phases[i] += d
    torch.cuda.synchronize()
    return (clock() - t0) / steps * 1e6


loop(20, True, True)
print(f"N={N} actor + step only:                      {loop(STEPS, False, False):8.1f} us/step")
print(f"N={N} + a bare stream synchronisation after every step (what ANY host read of the step's values costs this loop): "
      f"{loop(STEPS, False, False, bare_sync=True):8.1f} us/step")
print(f"N={N} + observed_components + executed actions: {loop(STEPS, True, False):8.1f} us/step")
print(f"N={N} + infos['episode'] .item() x22 (host copy): {loop(STEPS, True, True):8.1f} us/step")
print(f"N={N} + infos['episode'] .item() x22 (device tensors, as the reference returns them): {loop(STEPS, True, True, on_host=False):8.1f} us/step")
if "--breakdown" in sys.argv:
    ph = [0.0] * 5
    total = loop(STEPS, True, True, phases=ph)
    names = ("actor(states) [host: 8 launches]", "env.step(actions) [ois_step + logs launch + 4 output copies]",
             "observed_components x27 [waits for the published block]", "wheel_torqs / action_scale [1 launch]",
             "infos['episode'] items + .item() x22")
    print(f"host time per step of the full loop ({total:.1f} us/step):")
    for n_, p in zip(names, ph):
        print(f"   {p / STEPS * 1e6:7.1f} us  {n_}")
