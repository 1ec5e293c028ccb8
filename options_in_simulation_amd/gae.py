"""Generalized advantage estimation on the device (include/ois.h ``ois_gae``).

Drop-in for the nested ``compute_gae`` of ``PPO._update`` (src/my_agents/ppo.py:321-360) and
for ``PPOC.compute_advantages_and_returns`` (src/my_agents/ppoc.py:305-336): same arguments,
same return order, one HIP launch (two with the normalisation) instead of a Python loop over
the rollout with ~8 torch launches per step.
"""
from __future__ import annotations

from typing import Tuple

import torch

from . import _native

GAE_PPO, GAE_PPOC = 0, 1


def _run(mode: int, rewards: torch.Tensor, dones: torch.Tensor, values: torch.Tensor, last_values: torch.Tensor,
         discount_factor: float, lambda_coefficient: float, normalize: bool) -> Tuple[torch.Tensor, torch.Tensor]:
    L = _native.load()
    if rewards.device.type != "cuda":
        raise ValueError("ois_gae works on device tensors (there is no CPU fallback)")
    T = rewards.shape[0]
    if T == 0 or rewards.numel() == 0:
        raise ValueError("empty rollout")
    N = rewards.numel() // T
    r = rewards.contiguous().float()
    v = values.contiguous().float()
    lv = last_values.contiguous().float()
    d = dones.contiguous()
    d = d.view(torch.uint8) if d.dtype == torch.bool else d.to(torch.uint8)
    if v.numel() != T * N or d.numel() != T * N or lv.numel() != N:
        raise ValueError(f"shape mismatch: rewards {tuple(rewards.shape)}, values {tuple(values.shape)}, "
                         f"dones {tuple(dones.shape)}, last values {tuple(last_values.shape)}")
    returns, advantages = torch.empty_like(r), torch.empty_like(r)
    scratch, nbytes = None, 0
    if normalize:
        nbytes = L.ois_gae_scratch_bytes(N)
        scratch = torch.empty(nbytes // 8, dtype=torch.float64, device=r.device)
    with torch.cuda.device(r.device):
        stream = torch.cuda.current_stream().cuda_stream
        _native.check(L.ois_gae(mode, T, N, r.data_ptr(), d.data_ptr(), v.data_ptr(), lv.data_ptr(),
                                float(discount_factor), float(lambda_coefficient), int(normalize),
                                returns.data_ptr(), advantages.data_ptr(),
                                scratch.data_ptr() if scratch is not None else None, nbytes, stream))
    return returns, advantages


def compute_gae(rewards: torch.Tensor, dones: torch.Tensor, values: torch.Tensor, next_values: torch.Tensor,
                discount_factor: float = 0.99, lambda_coefficient: float = 0.95,
                normalize: bool = True) -> Tuple[torch.Tensor, torch.Tensor]:
    """ppo.py:321-360. ``next_values`` are the values of the states after the last rollout
    step (``last_values`` there). Returns ``(returns, advantages)``, advantages normalised by
    their mean and unbiased std unless ``normalize=False``."""
    return _run(GAE_PPO, rewards, dones, values, next_values, discount_factor, lambda_coefficient, normalize)


def compute_advantages_and_returns(rewards: torch.Tensor, terminated: torch.Tensor, values: torch.Tensor,
                                   last_values: torch.Tensor, gamma: float = 0.99,
                                   lam: float = 0.95) -> Tuple[torch.Tensor, torch.Tensor]:
    """ppoc.py:305-336 on explicit tensors. Returns ``(advantages, returns)`` (that order there)."""
    returns, advantages = _run(GAE_PPOC, rewards, terminated, values, last_values, gamma, lam, False)
    return advantages, returns
