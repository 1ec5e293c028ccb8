"""The data path of train/train_kaddpg_4d_nofifo_random_RLIL.py + SequentialTrainer
(src/my_trainers/base_KA_DDPG.py:153-293) on the MI355X environment, with the models and the
DDPG update left out: what changes in a train script, in one runnable file.

    python examples/kaddpg_loop.py --num-envs 65536 --timesteps 400 --logdir /tmp/ois_run
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from options_in_simulation_amd import DeviceFIFOMemory, VecEnv, default_task_cfg  # noqa: E402
from options_in_simulation_amd.tracking import Env0Tracker  # noqa: E402


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--num-envs", type=int, default=4096)
    ap.add_argument("--timesteps", type=int, default=200)
    ap.add_argument("--memory-size", type=int, default=16)
    ap.add_argument("--batch-size", type=int, default=4096)
    ap.add_argument("--logdir", default="/tmp/ois_kaddpg_run")
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args(argv)

    # --- train script :57-100: TASK_CFG patches, then the env (replaces get_env_instance + task + wrap_env)
    TASK_CFG = default_task_cfg()
    TASK_CFG["seed"] = args.seed
    TASK_CFG["task"]["env"]["numEnvs"] = args.num_envs
    env = VecEnv(TASK_CFG, device="cuda:0", variant="KA_DDPG_4D")
    device = env.device

    # --- :26-52 the actor's architecture (DeterministicMixin there); weights random here
    torch.manual_seed(args.seed)
    policy = torch.nn.Sequential(torch.nn.Linear(env.observation_space.shape[0], 512), torch.nn.ELU(),
                                 torch.nn.Linear(512, 512), torch.nn.ELU(), torch.nn.Linear(512, 128), torch.nn.ELU(),
                                 torch.nn.Linear(128, env.action_space.shape[0]), torch.nn.Tanh()).to(device)

    states, infos = env.reset()                                  # base_KA_DDPG.py:170
    memory = DeviceFIFOMemory(env, memory_size=args.memory_size)  # :208 FIFOMemory(...): after the first reset
    tracker = Env0Tracker(env, args.logdir, write_interval=max(1, args.timesteps // 4))
    t0 = time.perf_counter()
    for timestep in range(1, args.timesteps + 1):
        with torch.no_grad():
            actions = policy(states)                             # agent.act (:176)
        next_states, rewards, terminated, truncated, infos = env.step(actions)  # :182
        # record_transition (:205-224): nothing to do for the transition itself — the step kernel has
        # written states / executed actions / rewards / next_states / terminated into `memory`
        tracker.track(timestep, infos)                           # :184-262 tracking block
        if len(memory) >= args.batch_size:                       # ddpg.py _update: sample a mini-batch
            (s, a, r, ns, d), = memory.sample(["states", "actions", "rewards", "next_states", "terminated"], args.batch_size)
            assert s.shape == (args.batch_size, 6) and a.shape == (args.batch_size, 4)
        states = next_states                                     # :268-269 (num_envs > 1)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    tracker.close()
    st = env.task.native.stats()
    print(f"{args.timesteps} timesteps x {args.num_envs} arenas in {dt:.2f} s = {args.timesteps * args.num_envs / dt:.3e} env-steps/s; "
          f"{int(st.episodes)} episodes finished, mean return {st.return_sum / max(st.episodes, 1):.2f}; "
          f"events + CSV in {args.logdir}")
    env.close()
    return 0


if __name__ == "__main__":
    sys.exit(main())
