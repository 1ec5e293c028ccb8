"""Device-resident FIFO experience memory fed by the step kernel.

Mirrors the train scripts' ``FIFOMemory`` (train/train_kaddpg_4d_nofifo_random_RLIL.py:
105-206): tensors ``[memory_size, num_envs, dim]``, the row at ``memory_index`` is overwritten
and the index wraps, ``sample`` draws uniformly (without replacement by default) over the
filled part. The difference: nobody calls ``add_samples`` — ``ois_step`` / ``ois_rollout``
write the transition row themselves (states = previous observation, actions = executed
torques / action_scale as the trainer substitutes at src/my_trainers/base_KA_DDPG.py:212,
rewards, next_states, terminated), so a training loop needs no per-step host work for it.
"""
from __future__ import annotations

from typing import Dict, List, Sequence

import torch

from .vecenv import NativeEnv, VecEnv


class DeviceFIFOMemory:
    def __init__(self, env: "VecEnv | NativeEnv", memory_size: int, replacement: bool = False):
        native = env.task.native if isinstance(env, VecEnv) else env
        self.native = native
        self.memory_size = int(memory_size)
        self.num_envs = native.N
        self.device = native.device
        self.replacement = replacement
        self.tensors: Dict[str, torch.Tensor] = native.attach_replay(self.memory_size)
        self.sampling_indexes = None

    # skrl Memory surface used by the agents -------------------------------------------------
    @property
    def memory_index(self) -> int:
        return self.native.replay_cursor()[0]

    @property
    def filled(self) -> bool:
        return self.native.replay_cursor()[1]

    def __len__(self) -> int:
        idx, filled = self.native.replay_cursor()
        return self.memory_size * self.num_envs if filled else idx * self.num_envs

    def get_tensor_by_name(self, name: str, keepdim: bool = True) -> torch.Tensor:
        t = self.tensors[name]
        return t if keepdim else t.reshape(-1, t.shape[-1])

    def sample_by_index(self, names: Sequence[str], indexes: torch.Tensor, mini_batches: int = 1) -> List[List[torch.Tensor]]:
        flat = {n: self.tensors[n].reshape(-1, self.tensors[n].shape[-1]) for n in names}
        if mini_batches > 1:
            batches = torch.chunk(indexes, mini_batches)
            return [[flat[n][b] for n in names] for b in batches]
        return [[flat[n][indexes] for n in names]]

    def sample(self, names: Sequence[str], batch_size: int, mini_batches: int = 1) -> List[List[torch.Tensor]]:
        size = len(self)
        if size == 0:
            raise RuntimeError("the memory is empty")
        if self.replacement:
            indexes = torch.randint(0, size, (batch_size,), device=self.device)
        else:
            indexes = torch.randperm(size, dtype=torch.long, device=self.device)[:batch_size]
        self.sampling_indexes = indexes
        return self.sample_by_index(names, indexes, mini_batches)

    def close(self):
        self.native.detach_replay()
