"""Multi-GPU plumbing for the one exchange step the path has.

Arenas are independent (KAMMA.py:153, rl_task.py:187-193), so they shard by rank with no
data-path collective; the RNG is keyed by the GLOBAL arena id, which makes a shard
bit-identical to the same arenas inside a single-GPU run. The only communication is one
all-gather of a 64-byte statistics record per rank per logging interval (mirrors the
finished-episode tracking of src/my_agents/base.py:305-341 and the means of
KAMMA.py:480-488). Backend "nccl" is RCCL on ROCm; "gloo" is used by the CPU tests.
"""
from __future__ import annotations

from typing import Dict, List, Tuple

import torch
import torch.distributed as dist

from . import _abi

_FIELDS = ("episodes", "return_sum", "return_sqsum", "return_min", "return_max", "length_sum", "steps", "reserved")


def shard_range(total_arenas: int, rank: int, world: int) -> Tuple[int, int]:
    """[start, stop) of the global arena ids owned by `rank`: contiguous, balanced."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    base, rem = divmod(total_arenas, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def stats_to_tensor(st: _abi.OisStats, device) -> torch.Tensor:
    return torch.tensor([getattr(st, f) for f in _FIELDS], dtype=torch.float64, device=device)


def tensor_to_stats(t: torch.Tensor) -> _abi.OisStats:
    st = _abi.OisStats()
    for f, v in zip(_FIELDS, t.tolist()):
        setattr(st, f, v)
    return st


def gather_stats(st: _abi.OisStats, device=None, group=None) -> List[_abi.OisStats]:
    """ONE all-gather of the per-rank record; every rank gets the list in rank order."""
    if not (dist.is_available() and dist.is_initialized()):
        return [st]
    backend = dist.get_backend(group)
    dev = torch.device("cpu") if backend == "gloo" else (device or torch.device("cuda", torch.cuda.current_device()))
    mine = stats_to_tensor(st, dev)
    world = dist.get_world_size(group)
    out = torch.empty(world * mine.numel(), dtype=torch.float64, device=dev)  # flat: gloo and RCCL both accept it
    dist.all_gather_into_tensor(out, mine, group=group)
    return [tensor_to_stats(row) for row in out.view(world, -1).cpu()]


def merge_stats(parts: List[_abi.OisStats]) -> dict:
    """global mean / min / max / std of finished-episode returns (base.py:327-341)."""
    n = sum(p.episodes for p in parts)
    steps = sum(p.steps for p in parts)
    if n == 0:
        return {"episodes": 0.0, "steps": steps, "return_mean": None, "return_min": None, "return_max": None,
                "return_std": None, "length_mean": None}
    s1 = sum(p.return_sum for p in parts)
    s2 = sum(p.return_sqsum for p in parts)
    mean = s1 / n
    var = max(s2 / n - mean * mean, 0.0)
    return {"episodes": n, "steps": steps, "return_mean": mean,
            "return_min": min(p.return_min for p in parts if p.episodes),
            "return_max": max(p.return_max for p in parts if p.episodes),
            "return_std": var ** 0.5, "length_mean": sum(p.length_sum for p in parts) / n}


def gather_episode_info(info, num_envs: int, device=None, group=None, eval_keys: bool = True,
                        single_gamma: bool = False) -> Dict[str, float]:
    """Global ``extras["episode"]`` (KAMMA.py:480-488) from the per-rank vectors of
    ``ois_episode_info`` with ONE all-gather of 33 floats per rank: the per-key means of the
    most recent reset event are re-weighted by each rank's count of resetting arenas (ranks whose
    newest event is older than the global newest do not contribute), ``terrain_level`` by the
    ranks' arena counts, gamma is taken from a contributing rank. Keys and their order: ``_abi.episode_info_layout`` for
    the variant (``single_gamma``: the KA-DDPG files log one ``"gamma assist"``, KA_DDPG_4D.py:484; ``eval_keys``: the three
    sums of the eval files) — the same dict a single-rank ``infos["episode"]`` has."""
    vec = torch.as_tensor(info, dtype=torch.float64).flatten().cpu()
    mine = torch.cat([vec[:_abi.EPISODE_INFO_LEN], torch.tensor([float(num_envs)], dtype=torch.float64)])
    if dist.is_available() and dist.is_initialized():
        backend = dist.get_backend(group)
        dev = torch.device("cpu") if backend == "gloo" else (device or torch.device("cuda", torch.cuda.current_device()))
        world = dist.get_world_size(group)
        out = torch.empty(world * mine.numel(), dtype=torch.float64, device=dev)
        dist.all_gather_into_tensor(out, mine.to(dev), group=group)
        rows = out.view(world, -1).cpu()
    else:
        rows = mine.view(1, -1)
    step = rows[:, _abi.EI_STEP]
    cnt = torch.where(step == step.max(), rows[:, _abi.EI_COUNT], torch.zeros_like(step))
    total = float(cnt.sum())
    merged: Dict[str, float] = {}
    lead = int(torch.argmax(cnt))
    for key, slot in _abi.episode_info_layout(eval_keys, single_gamma):
        if slot in (16, 17):
            merged[key] = float(rows[lead, slot])
        elif slot == _abi.EI_TERRAIN_LEVEL:
            merged[key] = float((rows[:, slot] * rows[:, -1]).sum() / rows[:, -1].sum())
        elif slot == 31:
            merged[key] = 0.0
        else:
            merged[key] = float((rows[:, slot] * cnt).sum() / total) if total > 0 else float("nan")
    merged["count"] = total
    merged["step"] = float(step.max())
    return merged
