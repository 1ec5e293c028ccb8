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

    def __init__(self, variant, n, seed):
        self.co = Cosim(variant, n, seed)
        self.task = self.co.task
        self.num_envs, self.num_agents = n, 1
        self._first, self._obs, self._t = True, None, 0

    def _step(self, actions):
        out = self.co.step(self._t, actions.numpy().astype(np.float32))
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


def main():
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
