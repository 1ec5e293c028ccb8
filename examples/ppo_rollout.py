"""The data path of a PPO update as src/my_agents/ppo.py runs it (record_transition :229-283,
_update :313-380) on the MI355X environment: rollout memory written by the step kernel, the
agent's log_prob / values beside it, GAE by ois_gae, mini-batches by sample_all. The models are
stand-ins (random weights); the optimisation step is left out.

    python examples/ppo_rollout.py --num-envs 65536 --rollouts 16 --updates 5
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from options_in_simulation_amd import DeviceFIFOMemory, VecEnv, compute_gae  # noqa: E402


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--num-envs", type=int, default=4096)
    ap.add_argument("--rollouts", type=int, default=16)       # PPO_DEFAULT_CONFIG["rollouts"]
    ap.add_argument("--updates", type=int, default=3)
    ap.add_argument("--mini-batches", type=int, default=4)
    args = ap.parse_args(argv)

    env = VecEnv(num_envs=args.num_envs, device="cuda:0", variant="KAMMA_eval", seed=1)
    dev = env.device
    torch.manual_seed(0)
    policy = torch.nn.Sequential(torch.nn.Linear(6, 128), torch.nn.ELU(), torch.nn.Linear(128, 4), torch.nn.Sigmoid()).to(dev)
    value = torch.nn.Sequential(torch.nn.Linear(6, 128), torch.nn.ELU(), torch.nn.Linear(128, 1)).to(dev)

    states, _ = env.reset()
    # actions='policy': PPO stores the action it sampled, beside its log_prob (ppo.py:270-272), not the executed torque
    memory = DeviceFIFOMemory(env, memory_size=args.rollouts, actions="policy")
    for name in ("log_prob", "values", "returns", "advantages"):   # ppo.py:193-196
        memory.create_tensor(name, 1)
    names = ["states", "actions", "log_prob", "values", "returns", "advantages"]
    t0 = time.perf_counter()
    for update in range(args.updates):
        for _ in range(args.rollouts):
            with torch.no_grad():
                actions = policy(states)
                values = value(states)
                log_prob = -((actions - 0.5) ** 2).sum(1, keepdim=True)   # stand-in for the Gaussian log-density
            next_states, rewards, terminated, truncated, infos = env.step(actions)
            memory.add_samples(states=states, actions=actions, rewards=rewards, next_states=next_states,
                               terminated=terminated, truncated=truncated, log_prob=log_prob, values=values)  # :270-272
            states = next_states
        with torch.no_grad():
            last_values = value(states)                                    # :362-366
        returns, advantages = compute_gae(rewards=memory.get_tensor_by_name("rewards"),
                                          dones=memory.get_tensor_by_name("terminated"),
                                          values=memory.get_tensor_by_name("values"), next_values=last_values,
                                          discount_factor=0.99, lambda_coefficient=0.95)   # :370-375
        memory.set_tensor_by_name("returns", returns)
        memory.set_tensor_by_name("advantages", advantages)
        batches = memory.sample_all(names, mini_batches=args.mini_batches)  # :380
        assert len(batches) == args.mini_batches
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    n = args.updates * args.rollouts * args.num_envs
    print(f"{args.updates} updates x {args.rollouts} rollouts x {args.num_envs} arenas: {n / dt:.3e} env-steps/s incl. GAE and batching; "
          f"advantages mean {float(advantages.mean()):+.2e} std {float(advantages.std()):.3f}")
    env.close()
    return 0


if __name__ == "__main__":
    sys.exit(main())
