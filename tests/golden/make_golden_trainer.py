#!/usr/bin/env python3
"""The CALLER's side of the boundary, executed: the reference's own trainer loop driven against a recording environment
(build container only; needs /root/reference). Writes tests/golden/refexec_trainer_trace.json.

`Trainer.single_agent_train` and `Trainer.single_agent_eval` (src/my_trainers/base_KA_DDPG.py:153-293, :295-350) are taken
out of the reference's file by ast and run unchanged (they need only torch, tqdm and sys). `self.env` is a recording proxy:
every attribute read, call, argument and dictionary key the trainer touches on the environment is logged with the type /
shape / dtype of what came back, in order. Behind the proxy sits the reference's OWN task (refexec_task.py: the statements
of TorqueDistributionTask that do not need Isaac Sim, stepped on the oracle's physics — make_golden_cosim.Cosim) inside the
third-party wrapper's published semantics (skrl's omniverse-isaacgym wrapper: reset() steps once with zero actions the
first time and returns the cached observation afterwards; step() returns obs, rew.view(-1, 1), reset_buf.view(-1, 1),
zeros, extras — SURVEY.md 8b), so the shapes and dtypes in the trace are those of the reference's own buffers.
`self.agents` records what reaches the agent (record_transition's arguments, the track_data tags).

Only names, shapes, dtypes and key lists are stored — the trace is data; the reference's Python stays here.
tests/test_gpu_env_api.py::test_trainer_trace_replays_on_the_device replays it against the product's VecEnv on the GPU.

    python tests/golden/make_golden_trainer.py
"""
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

from make_golden_cosim import Cosim  # noqa: E402
from make_golden_ref import extract  # noqa: E402


def summary(v):
    if isinstance(v, torch.Tensor):
        return {"type": "tensor", "shape": list(v.shape), "dtype": str(v.dtype).replace("torch.", "")}
    if isinstance(v, bool):
        return {"type": "bool"}
    if isinstance(v, int):
        return {"type": "int"}
    if isinstance(v, float):
        return {"type": "float"}
    if isinstance(v, dict):
        return {"type": "dict", "keys": sorted(v), "values": {k: summary(v[k]) for k in sorted(v)}}
    if isinstance(v, (tuple, list)):
        return {"type": "tuple", "items": [summary(x) for x in v]}
    if v is None:
        return {"type": "none"}
    return {"type": type(v).__name__}


class Rec:
    """logs attribute reads / calls / item reads of the object behind it; tensors and scalars pass through as they are"""

    def __init__(self, obj, path, log):
        object.__setattr__(self, "_o", obj); object.__setattr__(self, "_p", path); object.__setattr__(self, "_log", log)

    def __getattr__(self, name):
        v = getattr(self._o, name)
        p = f"{self._p}.{name}"
        if callable(v) and not isinstance(v, torch.Tensor):
            def call(*a, **k):
                r = v(*a, **k)
                self._log.append({"op": "call", "path": p, "args": [summary(x) for x in a], "result": summary(r)})
                return r
            return call
        if isinstance(v, dict) or hasattr(v, "__dict__") and not isinstance(v, (torch.Tensor, int, float, str)):
            self._log.append({"op": "getattr", "path": p, "result": {"type": type(v).__name__ if not isinstance(v, dict) else "dict"}})
            return Rec(v, p, self._log)
        self._log.append({"op": "getattr", "path": p, "result": dict(summary(v), **({"value": v} if name in ("num_envs", "num_agents") else {}))})
        return v

    def __getitem__(self, key):
        v = self._o[key]
        self._log.append({"op": "getitem", "path": self._p, "key": key, "result": summary(v)})
        return v


class WrappedRefEnv:
    """skrl's wrapper semantics (third-party, restated from its published source) around the reference's own
    task statements on the oracle's physics"""

    def __init__(self, variant, n, seed, events=None):
        self.events = events or {}
        self.co = Cosim(variant, n, seed)
        self.task = self.co.task
        self.num_envs, self.num_agents = n, 1
        self._first, self._obs, self._t = True, None, 0
        self.history = []

    def _step(self, actions):
        out = self.co.step(self._t, actions.numpy().astype(np.float32), self.events.get(self._t, ()))
        self.history.append(out)
        self._t += 1
        T = self.task
        obs = torch.clamp(T.obs_buf, -T.clip_obs, T.clip_obs).clone()           # rl_task.py:99 / VecEnvBase.step
        return obs, T.rew_buf.clone(), T.reset_buf.clone(), dict(T.extras)

    def reset(self):
        if self._first:
            self.co.start(7)
            self._obs, _, _, _ = self._step(torch.zeros(self.num_envs, self.co.cfg.action_dim))
            self._first = False
        return self._obs, {}

    def step(self, actions):
        obs, rew, reset, extras = self._step(actions)
        self._obs = obs
        terminated = reset.view(-1, 1)
        return obs, rew.view(-1, 1), terminated, torch.zeros_like(terminated), extras

    def render(self, *a, **k):
        return None


class RecAgent:
    def __init__(self, log, action_dim, lo):
        self.log, self.A, self.lo = log, action_dim, lo
        self.g = torch.Generator().manual_seed(0)

    def pre_interaction(self, timestep, timesteps):
        pass

    def post_interaction(self, timestep, timesteps):
        self.log.append({"op": "agent", "name": "post_interaction"})

    def act(self, states, timestep, timesteps):
        self.log.append({"op": "agent", "name": "act", "states": summary(states)})
        a = (1.0 - self.lo) * torch.rand((states.shape[0], self.A), generator=self.g) + self.lo
        return a, None, {}

    def record_transition(self, **kw):
        self.log.append({"op": "agent", "name": "record_transition", "kwargs": {k: summary(v) for k, v in kw.items()}})

    def track_data(self, tag, value):
        self.log.append({"op": "agent", "name": "track_data", "tag": tag, "value": summary(value)})


def run(method, variant, n, timesteps, headless, seed):
    fn, _ = extract("my_trainers/base_KA_DDPG.py", "Trainer", method,
                    extra_globals={"tqdm": __import__("tqdm"), "sys": sys})
    log = []
    env = WrappedRefEnv(variant, n, seed)
    lo = -1.0 if variant.startswith("KA_DDPG") else 0.0
    # single_agent_eval ends a step with super(type(self.agents), self.agents).post_interaction: the agent needs a base class
    base = type("AgentBase", (), {"post_interaction": lambda self, timestep, timesteps: log.append({"op": "agent", "name": "base.post_interaction"})})
    agent_cls = type("RecAgentWithBase", (RecAgent, base), {})
    S = types.SimpleNamespace(env=Rec(env, "env", log), agents=agent_cls(log, env.co.cfg.action_dim, lo), num_simultaneous_agents=1,
                              initial_timestep=0, timesteps=timesteps, disable_progressbar=True, headless=headless, environment_info="episode")
    fn(S)
    # the num_envs == 1 branch: after a terminated step the trainer calls env.reset() again; the wrapper hands back its cached
    # observation, i.e. the observation of the step that just ended the episode
    resets = [i for i, e in enumerate(log) if e["op"] == "call" and e["path"] == "env.reset"]
    return {"method": method, "variant": variant, "num_envs": n, "timesteps": timesteps, "headless": headless,
            "env_reset_calls": len(resets), "events": log}


def extract_method_in_class(relpath, cls, method, base, extra_globals):
    """a method that uses zero-argument super(): compiled inside a class statement (so that the __class__ cell exists) whose
    base is `base`; returns the class"""
    import ast
    from make_golden_ref import REF
    path = os.path.join(REF, relpath)
    tree = ast.parse(open(path).read(), filename=path)
    c = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == cls)
    fn = next(n for n in c.body if isinstance(n, ast.FunctionDef) and n.name == method)
    new = ast.ClassDef(name=cls, bases=[ast.Name("Base_", ast.Load())], keywords=[], body=[fn], decorator_list=[])
    mod = ast.Module(body=[new], type_ignores=[])
    ast.fix_missing_locations(mod)
    ns = {"torch": torch, "np": np, "Base_": base}
    ns.update(extra_globals)
    exec(compile(mod, path, "exec"), ns)
    return ns[cls]


def fifo_case(n=3, timesteps=41, memory_size=16, seed=31):
    """The whole caller chain as the reference wrote it: Trainer.single_agent_train (base_KA_DDPG.py:153) ->
    DDPG.record_transition (my_agents/ddpg.py:270: super().record_transition = Agent.record_transition, base.py:270, then
    self.memory.add_samples) -> FIFOMemory.add_samples (train/train_kaddpg_4d_nofifo_random_RLIL.py:152, the script's own
    class), all executed unchanged on the reference's own task statements. skrl's Memory base class (third-party) is not used:
    what its constructor and create_tensor leave on the object — tensors[name] of shape (memory_size, num_envs, size),
    memory_index, env_index, filled — is set here as its published source does. Output: the memory after `timesteps` steps
    (it wraps) plus the run's inputs, so that the oracle's transition ring (orc_replay_*, the mirror of ois_replay_*) can
    replay it."""
    import collections
    from typing import Any, List, Tuple
    train, _ = extract("my_trainers/base_KA_DDPG.py", "Trainer", "single_agent_train", extra_globals={"tqdm": __import__("tqdm"), "sys": sys})
    AgentBase = extract_method_in_class("my_agents/base.py", "Agent", "record_transition", object, {"Any": Any})
    DDPG = extract_method_in_class("my_agents/ddpg.py", "DDPG", "record_transition", AgentBase, {"Any": Any})
    add_samples, _ = extract("../train/train_kaddpg_4d_nofifo_random_RLIL.py", "FIFOMemory", "add_samples",
                             extra_globals={"Tuple": Tuple, "List": List})
    log = []
    env = WrappedRefEnv("KA_DDPG_4D", n, seed, events={33: [("fall", 1)], 34: [("fall", 1)]})   # an episode ends inside the last 16 rows

    class Memory:   # what skrl's Memory.__init__ / create_tensor leave behind (shapes per its published source)
        pass
    mem = Memory()
    mem.memory_size, mem.num_envs, mem.memory_index, mem.env_index, mem.filled = memory_size, n, 0, 0, False
    mem.tensors = {"states": torch.zeros(memory_size, n, 6), "actions": torch.zeros(memory_size, n, 4), "rewards": torch.zeros(memory_size, n, 1),
                   "next_states": torch.zeros(memory_size, n, 6), "terminated": torch.zeros(memory_size, n, 1, dtype=torch.bool)}
    mem.add_samples = types.MethodType(add_samples, mem)

    class Agent(DDPG):
        def __init__(self):
            self.write_interval, self._cumulative_rewards, self._cumulative_timesteps = 800, None, None
            self._track_rewards, self._track_timesteps = collections.deque(maxlen=100), collections.deque(maxlen=100)
            self.tracking_data = collections.defaultdict(list)
            self.memory, self._rewards_shaper, self.secondary_memories = mem, None, []
            self.g = torch.Generator().manual_seed(1)

        def pre_interaction(self, timestep, timesteps): pass
        def post_interaction(self, timestep, timesteps): pass
        def track_data(self, tag, value): pass

        def act(self, states, timestep, timesteps):
            return 2.0 * torch.rand((states.shape[0], 4), generator=self.g) - 1.0, None, {}
    S = types.SimpleNamespace(env=env, agents=Agent(), num_simultaneous_agents=1, initial_timestep=0, timesteps=timesteps,
                              disable_progressbar=True, headless=True, environment_info="episode")
    train(S)
    h = env.history   # [0] = the zero-action step inside the first env.reset()
    out = {"first_reset_u": env.co.first_reset_u, "memory_size": np.int32(memory_size), "timesteps": np.int32(timesteps), "seed": np.int32(seed),
           "memory_index": np.int32(mem.memory_index), "filled": np.bool_(mem.filled)}
    for k in ("actions", "switch_u", "reset_u", "physics"):
        out["step__" + k] = np.stack([x[k] for x in h])
    for k, v in mem.tensors.items():
        out["memory__" + k] = v.numpy()
    out["track_rewards"] = np.array(S.agents._track_rewards, np.float64)
    return out


def main():
    np.savez_compressed(os.path.join(HERE, "refexec_fifo.npz"), **fifo_case())
    runs = [run("single_agent_train", "KAMMA_medium", 4, 3, False, 21),      # num_envs > 1, rendering on
            run("single_agent_train", "KAMMA_medium", 1, 101, True, 22),      # num_envs == 1 across the timeout at step 99
            run("single_agent_train", "KA_DDPG_2D", 3, 2, True, 23),
            run("single_agent_eval", "KAMMA_eval", 2, 2, True, 24)]
    # collapse the long single-arena run: identical consecutive steps are stored once with a repeat count
    for r in runs:
        steps, cur = [], []
        for e in r["events"]:
            cur.append(e)
            if e["op"] == "agent" and e["name"] in ("post_interaction", "base.post_interaction") and cur and \
                    sum(1 for x in cur if x["op"] == "agent" and x["name"] in ("post_interaction", "base.post_interaction")) == (2 if r["method"] == "single_agent_train" else 1):
                steps.append(cur); cur = []
        tail = cur
        packed = []
        for s in steps:
            if packed and json.dumps(packed[-1]["events"], sort_keys=True) == json.dumps(s, sort_keys=True):
                packed[-1]["repeat"] += 1
            else:
                packed.append({"repeat": 1, "events": s})
        r["events"] = None
        r["blocks"] = packed
        r["tail"] = tail
        del r["events"]
    path = os.path.join(HERE, "refexec_trainer_trace.json")
    json.dump({"source": "src/my_trainers/base_KA_DDPG.py Trainer.single_agent_train :153-293, single_agent_eval :295-350, executed",
               "runs": runs}, open(path, "w"), indent=0)
    print("wrote", path, os.path.getsize(path), "bytes;", [(r["variant"], r["num_envs"], r["env_reset_calls"], len(r["blocks"])) for r in runs])


if __name__ == "__main__":
    main()
