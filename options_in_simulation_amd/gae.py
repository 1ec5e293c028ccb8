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
_SCRATCH = {}  # (device index, stream, arenas) -> moments scratch of ois_gae_scratch_bytes()


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
    # one allocation for both outputs; the moments scratch is kept per (device, stream, N): at the reference's rollout lengths
    # the launches take ~10 us and every allocation of the wrapper costs a third of that on the host
    both = torch.empty((2,) + tuple(r.shape), dtype=torch.float32, device=r.device)
    returns, advantages = both[0], both[1]
    stream = torch.cuda.current_stream(r.device).cuda_stream
    scratch, nbytes = None, 0
    if normalize:
        key = (r.device.index, stream, N)
        scratch = _SCRATCH.get(key)
        if scratch is None:
            if len(_SCRATCH) > 16:
                _SCRATCH.clear()
            scratch = _SCRATCH[key] = torch.empty(L.ois_gae_scratch_bytes(N) // 8, dtype=torch.float64, device=r.device)
        nbytes = scratch.numel() * 8
    def launch():
        _native.check(L.ois_gae(mode, T, N, r.data_ptr(), d.data_ptr(), v.data_ptr(), lv.data_ptr(),
                                float(discount_factor), float(lambda_coefficient), int(normalize),
                                returns.data_ptr(), advantages.data_ptr(),
                                scratch.data_ptr() if scratch is not None else None, nbytes, stream))
    if torch.cuda.current_device() == r.device.index:
        launch()
    else:  # ois_gae launches on the calling thread's current device: the tensors' device for the call
        with torch.cuda.device(r.device):
            launch()
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
