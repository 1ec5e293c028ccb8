"""Device-resident FIFO experience memory fed by the step kernel.

Mirrors the train scripts' ``FIFOMemory`` (train/train_kaddpg_4d_nofifo_random_RLIL.py:
105-206): tensors ``[memory_size, num_envs, dim]``, the row at ``memory_index`` is overwritten
and the index wraps, ``sample`` draws uniformly (without replacement by default) over the
filled part. The difference: nobody calls ``add_samples`` for the transition itself —
``ois_step`` / ``ois_rollout`` write the row (states = previous observation, actions = executed
torques / action_scale as the trainer substitutes at src/my_trainers/base_KA_DDPG.py:212,
rewards, next_states, terminated), so a training loop needs no per-step host work for it.

Which actions a row holds (``actions=`` of the constructor):

* ``"executed"`` — the kernel-written ``wheel_torqs / action_scale`` [4]: what
  ``base_KA_DDPG.py:212`` records, i.e. the trainer of the KAMMA and ``kaddpg_4d`` scripts. Whatever
  ``add_samples(actions=...)`` is handed is ignored, as that trainer overwrites it too.
* ``"policy"`` — the agent's own action, width ``action_dim``: what ``base_KA`` (train_kaddpg_1d / 2d
  via ``sequential_KA``) and PPO's ``record_transition`` (ppo.py:270-272, the sampled action whose
  ``log_prob`` sits beside it) store. When gamma1 > 0 or the clip is active the executed torque is
  not the policy's action, and a PPO ratio or a 1-D / 2-D critic trained on it would be wrong, so
  here ``add_samples(actions=...)`` copies the passed actions into the row; the executed ones stay
  available as ``executed_actions``. Default for ``action_dim != 4`` (the executed width is always 4).

The on-policy agents keep more per step (``log_prob``, ``values``, later ``returns`` and
``advantages``: src/my_agents/ppo.py:189-196): ``create_tensor`` adds such tensors beside the
kernel-written ones and ``add_samples(**tensors)`` puts the agent's values into the row the
kernel has just written, so the rollout memory of a PPO update lives on the device end to end
(``gae.compute_gae`` consumes ``get_tensor_by_name("rewards")`` etc. as they are).
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import torch

from .vecenv import NativeEnv, VecEnv

KERNEL_TENSORS = ("states", "actions", "rewards", "next_states", "terminated")


def indices_without_replacement(size: int, batch_size: int, device, generator=None) -> torch.Tensor:
    """``torch.randperm(size)[:batch_size]`` in distribution (a uniformly random ``batch_size``-subset
    in uniformly random order) without permuting the whole memory — the reference's FIFOMemory
    (``train_kaddpg_4d_nofifo_random_RLIL.py:139``) holds 35 000 samples of one arena, the same
    35 000 rows of 65 536 arenas are 2.3 G samples, whose permutation would be 18 GB per call.

    The first ``batch_size`` DISTINCT values of an i.i.d. uniform stream are exactly such a
    subset. ``2 * batch_size`` candidates are drawn, later duplicates are masked out (stable sort
    by value, compare neighbours), the survivors are compacted in stream order with a prefix sum;
    all shapes are fixed, nothing synchronises. It is used when ``8 * batch_size <= size``: then
    fewer than ``batch_size`` distinct candidates among ``2 * batch_size`` draws would need more
    than half of them to collide with at most 1/4 occupancy — no realistic probability — and the
    buffer is pre-filled with the leading candidates so that even then the result is in range.
    Denser requests fall back to ``randperm``."""
    if batch_size > size:
        raise ValueError(f"cannot draw {batch_size} distinct samples out of {size}")
    if 8 * batch_size > size:
        return torch.randperm(size, dtype=torch.long, device=device, generator=generator)[:batch_size]
    m = 2 * batch_size
    cand = torch.randint(0, size, (m,), device=device, generator=generator)
    srt, order = torch.sort(cand, stable=True)          # equal values keep their stream order
    dup_sorted = torch.zeros(m, dtype=torch.bool, device=device)
    dup_sorted[1:] = srt[1:] == srt[:-1]                # every occurrence after the first
    first = torch.ones(m, dtype=torch.bool, device=device)
    first[order] = ~dup_sorted                          # back to stream positions
    rank = torch.cumsum(first, 0) - 1                   # index among the distinct values, in stream order
    target = torch.where(first & (rank < batch_size), rank, torch.full_like(rank, batch_size))
    out = torch.empty(batch_size + 1, dtype=torch.long, device=device)
    out[:batch_size] = cand[:batch_size]
    out.scatter_(0, target, cand)                       # slot batch_size collects everything unused
    return out[:batch_size]


class DeviceFIFOMemory:
    def __init__(self, env: "VecEnv | NativeEnv", memory_size: int, replacement: bool = False,
                 actions: Optional[str] = None):
        native = env.task.native if isinstance(env, VecEnv) else env
        self.native = native
        self.memory_size = int(memory_size)
        self.num_envs = native.N
        self.device = native.device
        self.replacement = replacement
        self.tensors: Dict[str, torch.Tensor] = native.attach_replay(self.memory_size)
        self.sampling_indexes = None
        if actions is None:
            actions = "executed" if native.A == 4 else "policy"
        if actions not in ("executed", "policy"):
            raise ValueError("actions must be 'executed' or 'policy'")
        if actions == "executed" and native.A != 4:
            raise ValueError(f"executed actions are the 4 wheel torques; this task's action_dim is {native.A}: use actions='policy'")
        self.actions_mode = actions
        self._kernel_tensors = KERNEL_TENSORS
        if actions == "policy":
            # the kernel keeps writing the executed torques; the row's `actions` becomes the agent's own
            self.tensors["executed_actions"] = self.tensors["actions"]
            self.tensors["actions"] = torch.zeros((self.memory_size, self.num_envs, native.A), dtype=torch.float32, device=self.device)
            self._kernel_tensors = tuple(n for n in KERNEL_TENSORS if n != "actions")

    # ---- skrl Memory surface used by the agents ---------------------------------------------
    @property
    def memory_index(self) -> int:
        return self.native.replay_cursor()[0]

    @property
    def filled(self) -> bool:
        return self.native.replay_cursor()[1]

    def __len__(self) -> int:
        idx, filled = self.native.replay_cursor()
        return self.memory_size * self.num_envs if filled else idx * self.num_envs

    def get_tensor_names(self):
        return sorted(self.tensors)

    def create_tensor(self, name: str, size: int = 1, dtype: Optional[torch.dtype] = None, keep_dimensions: bool = False) -> bool:
        """skrl ``Memory.create_tensor``: a ``[memory_size, num_envs, size]`` tensor beside the
        kernel-written ones (``log_prob``, ``values``, ``returns``, ``advantages`` …). The five
        transition tensors exist already; asking for one of them only checks its width."""
        if name in self.tensors:
            if (name in KERNEL_TENSORS or name == "executed_actions") and self.tensors[name].shape[-1] != int(size):
                raise ValueError(f"{name} exists with width {self.tensors[name].shape[-1]}, not {size}"
                                 + (" (the executed wheel torques; actions='policy' stores the agent's own)" if name == "actions" else ""))
            return False
        self.tensors[name] = torch.zeros((self.memory_size, self.num_envs, int(size)), dtype=dtype or torch.float32,
                                         device=self.device)
        return True

    def add_samples(self, **tensors: torch.Tensor) -> None:
        """Complete the row the step kernel has just written (call after ``env.step``) with the
        agent's own per-step tensors. Transition tensors the kernel wrote (states, rewards, …, and the
        actions in ``"executed"`` mode), passed along as ``record_transition`` does (ppo.py:270-272),
        are ignored; in ``"policy"`` mode the passed ``actions`` are stored."""
        idx, _ = self.native.replay_cursor()
        row = (idx - 1) % self.memory_size
        for name, value in tensors.items():
            if name in self._kernel_tensors or name in ("truncated", "executed_actions"):
                continue
            if name not in self.tensors:
                raise KeyError(f"unknown memory tensor {name!r}: create_tensor() it first")
            self.tensors[name][row].copy_(value.reshape(self.num_envs, -1))

    def get_tensor_by_name(self, name: str, keepdim: bool = True) -> torch.Tensor:
        t = self.tensors[name]
        return t if keepdim else t.reshape(-1, t.shape[-1])

    def set_tensor_by_name(self, name: str, tensor: torch.Tensor) -> None:
        self.tensors[name].copy_(tensor.reshape(self.tensors[name].shape))

    def _gather(self, names: Sequence[str], indexes: torch.Tensor) -> List[torch.Tensor]:
        """the kernel-written tensors in one launch (ois_replay_sample), the agent's own by indexing"""
        fused = [n for n in names if n in self._kernel_tensors]
        got = self.native.replay_sample(indexes, fused) if fused else {}
        for n in names:
            if n not in got:
                t = self.tensors[n]
                got[n] = t.reshape(-1, t.shape[-1])[indexes]
        return [got[n] for n in names]

    def sample_by_index(self, names: Sequence[str], indexes: torch.Tensor, mini_batches: int = 1) -> List[List[torch.Tensor]]:
        if mini_batches > 1:
            return [self._gather(names, b) for b in torch.chunk(indexes, mini_batches)]
        return [self._gather(names, indexes)]

    def sample(self, names: Sequence[str], batch_size: int, mini_batches: int = 1) -> List[List[torch.Tensor]]:
        """FIFOMemory.sample (train_kaddpg_4d_nofifo_random_RLIL.py:112-145): ``batch_size`` indexes
        with or without replacement (the scripts use ``replacement=False``), then the gather — one
        launch here."""
        size = len(self)
        if size == 0:
            raise RuntimeError("the memory is empty")
        if self.replacement:
            indexes = torch.randint(0, size, (batch_size,), device=self.device)
        else:
            indexes = indices_without_replacement(size, batch_size, self.device)
        self.sampling_indexes = indexes
        return self.sample_by_index(names, indexes, mini_batches)

    def sample_all(self, names: Sequence[str], mini_batches: int = 1) -> List[List[torch.Tensor]]:
        """every stored sample once, in random order, split into mini-batches (what
        ``PPO._update`` iterates over, ppo.py:380)"""
        size = self.memory_size * self.num_envs
        indexes = torch.randperm(size, dtype=torch.long, device=self.device)
        return self.sample_by_index(names, indexes, mini_batches)

    def close(self):
        self.native.detach_replay()
