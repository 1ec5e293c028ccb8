"""N>1 path on CPU: world_size-2 gloo. Arena sharding + the one all-gather of episode
statistics per logging interval. The oracle stands in for the per-rank
environment (the HIP env needs a GPU; its sharding is covered by -m gpu tests)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from options_in_simulation_amd import _abi, build_config
from options_in_simulation_amd.dist import gather_episode_info, gather_stats, merge_stats, shard_range


def test_shard_range_partitions():
    for total, world in ((524288, 8), (10, 3), (7, 8), (65536, 1)):
        spans = [shard_range(total, r, world) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == total
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        sizes = [b - a for a, b in spans]
        assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_range(10, 3, 3)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, total, steps, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle.binding import OracleEnv
    lo, hi = shard_range(total, rank, world)
    env = OracleEnv(build_config(variant="KAMMA_medium", num_envs=hi - lo, seed=11, arena_offset=lo, max_sim_steps=80.0))
    env.reset(11)
    env.rollout(steps, _abi.POLICY_RANDOM)
    parts = gather_stats(env.stats(reset_after=True))
    assert len(parts) == world
    merged = merge_stats(parts)
    np.save(os.path.join(out_dir, f"state_{rank}.npy"), env.buffer(_abi.BUF_POSE))
    info = gather_episode_info(env.episode_info(), hi - lo)
    if rank == 0:
        np.save(os.path.join(out_dir, "info.npy"), np.array([info[k] for k in sorted(info)]))
        np.save(os.path.join(out_dir, "merged.npy"), np.array([merged["episodes"], merged["return_mean"], merged["return_min"],
                                                                 merged["return_max"], merged["steps"], merged["length_mean"]]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharded_run_matches_single_process(tmp_path):
    total, steps, world = 100, 110, 2
    mp.spawn(_worker, args=(world, _free_port(), total, steps, str(tmp_path)), nprocs=world, join=True)
    from oracle.binding import OracleEnv
    full = OracleEnv(build_config(variant="KAMMA_medium", num_envs=total, seed=11, max_sim_steps=80.0))
    full.reset(11)
    full.rollout(steps, _abi.POLICY_RANDOM)
    cat = np.concatenate([np.load(tmp_path / f"state_{r}.npy") for r in range(world)])
    assert np.array_equal(cat, full.buffer(_abi.BUF_POSE))  # sharding does not change any arena
    merged = np.load(tmp_path / "merged.npy")
    st = full.stats()
    assert merged[0] == st.episodes == total
    assert merged[1] == pytest.approx(st.return_sum / st.episodes, rel=1e-5)
    assert merged[2] == pytest.approx(st.return_min, rel=1e-6) and merged[3] == pytest.approx(st.return_max, rel=1e-6)
    assert merged[4] == total * steps and merged[5] == 99.0
    # the merged extras of the two shards = the extras of the unsharded run (same reset event)
    ei = full.episode_info()
    ref = {key: float(ei[slot]) for slot, key in _abi.EPISODE_INFO_KEYS.items()}
    ref["count"], ref["step"] = float(ei[_abi.EI_COUNT]), float(ei[_abi.EI_STEP])
    for slot, key in _abi.EPISODE_INFO_EVAL_KEYS.items():
        ref[key] = float(ei[slot])
    got = dict(zip(sorted(ref), np.load(tmp_path / "info.npy")))
    assert got["count"] == ref["count"] > 0 and got["step"] == ref["step"]
    for key in ref:
        assert got[key] == pytest.approx(ref[key], rel=1e-5, abs=1e-6, nan_ok=True), key


def test_gather_without_process_group_is_identity():
    st = _abi.OisStats()
    st.episodes, st.return_sum = 3.0, -6.0
    assert gather_stats(st)[0].episodes == 3.0
    m = merge_stats([st])
    assert m["return_mean"] == -2.0
    assert merge_stats([_abi.OisStats()])["return_mean"] is None
    vec = np.zeros(_abi.EPISODE_INFO_LEN); vec[_abi.EI_COUNT], vec[_abi.EI_STEP], vec[8], vec[16] = 4, 99, -2.5, 0.75
    one = gather_episode_info(vec, 16)
    assert one["rew_Final reward"] == -2.5 and one["gamma_1 action assist"] == 0.75 and one["count"] == 4 and one["step"] == 99


def test_merge_stats_std_at_half_a_million_episodes():
    """the finished-episode sums are accumulated in double (oracle and library alike) so
    that the variance — a difference of two large numbers — survives 524 288 episodes (BASELINE.json
    configs[4]'s count after one timeout). The oracle env at 8 192 arenas x 64 timeouts has that many."""
    import numpy as np
    from options_in_simulation_amd import _abi, build_config
    from options_in_simulation_amd.dist import merge_stats
    from oracle.binding import OracleEnv
    N, rounds = 2048, 4
    env = OracleEnv(build_config(variant="KAMMA_medium", num_envs=N, seed=1, decimation=1, extra_substeps=0), threads=8)
    env.reset(1)
    returns = []
    for r in range(rounds * 99):
        before = env.buffer(_abi.BUF_SUMS_B)[:, 3].copy()
        env.rollout(1, _abi.POLICY_RANDOM)
        d = env.done.astype(bool)
        if d.any():
            # the finished episode's return = its Final-reward sum incl. this step's reward
            returns.append((before + env.rew)[d].astype(np.float64))
    st = env.stats()
    x = np.concatenate(returns)
    assert st.episodes == len(x) == N * rounds
    # scale the same statistics to 524 288 episodes by merging 64 copies of the record (what 64 ranks would gather)
    m = merge_stats([st] * 64)
    assert m["episodes"] == 524288
    ref_mean, ref_std = x.mean(), x.std()
    assert m["return_mean"] == pytest.approx(ref_mean, rel=1e-6)
    assert m["return_std"] == pytest.approx(ref_std, rel=1e-4)
    assert st.return_min == pytest.approx(x.min(), rel=1e-6) and st.return_max == pytest.approx(x.max(), rel=1e-6)


def _product_worker(rank, world, port, total, steps, out_dir):
    """one PRODUCT environment shard per rank (the HIP path; the ranks share the box's one GPU, which gloo
    allows and RCCL does not), the statistics and the extras through the real collectives"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from options_in_simulation_amd.vecenv import NativeEnv
    lo, hi = shard_range(total, rank, world)
    env = NativeEnv(build_config(variant="KAMMA_medium", num_envs=hi - lo, seed=11, arena_offset=lo, max_sim_steps=80.0), "cuda:0")
    env.reset(11)
    env.rollout(steps, _abi.POLICY_RANDOM)
    parts = gather_stats(env.stats(reset_after=True))
    merged = merge_stats(parts)
    np.save(os.path.join(out_dir, f"state_{rank}.npy"), env.buffer(_abi.BUF_POSE).cpu().numpy())
    info = gather_episode_info(env.episode_info().cpu().numpy(), hi - lo)
    if rank == 0:
        np.save(os.path.join(out_dir, "info.npy"), np.array([info[k] for k in sorted(info)]))
        np.save(os.path.join(out_dir, "merged.npy"), np.array([merged["episodes"], merged["return_mean"], merged["return_std"], merged["return_min"],
                                                                 merged["return_max"], merged["steps"], merged["length_mean"]]))
    dist.barrier()
    dist.destroy_process_group()
    env.close()


@pytest.mark.gpu
def test_two_product_shards_on_gloo_match_the_unsharded_product_run(tmp_path):
    """the multi-rank rehearsals used the oracle as the rank environment. Here two processes
    each run a shard on the HIP path (arena_offset = shard start), gather through gloo, and the parent
    compares with ONE product environment over all arenas: identical arenas, equal merged statistics."""
    from options_in_simulation_amd.vecenv import NativeEnv
    total, steps, world = 4000, 205, 2
    mp.spawn(_product_worker, args=(world, _free_port(), total, steps, str(tmp_path)), nprocs=world, join=True)
    full = NativeEnv(build_config(variant="KAMMA_medium", num_envs=total, seed=11, max_sim_steps=80.0), "cuda:0")
    full.reset(11)
    full.rollout(steps, _abi.POLICY_RANDOM)
    cat = np.concatenate([np.load(tmp_path / f"state_{r}.npy") for r in range(world)])
    assert np.array_equal(cat, full.buffer(_abi.BUF_POSE).cpu().numpy())
    merged = np.load(tmp_path / "merged.npy")
    st = full.stats()
    assert merged[0] == st.episodes == 2 * total
    mean = st.return_sum / st.episodes
    assert merged[1] == pytest.approx(mean, rel=1e-12)
    assert merged[2] == pytest.approx((st.return_sqsum / st.episodes - mean * mean) ** 0.5, rel=1e-9)
    assert merged[3] == st.return_min and merged[4] == st.return_max
    assert merged[5] == total * steps and merged[6] == 99.0
    ei = full.episode_info().cpu().numpy()
    ref = {key: float(ei[slot]) for slot, key in _abi.EPISODE_INFO_KEYS.items()}
    ref["count"], ref["step"] = float(ei[_abi.EI_COUNT]), float(ei[_abi.EI_STEP])
    for slot, key in _abi.EPISODE_INFO_EVAL_KEYS.items():
        ref[key] = float(ei[slot])
    got = dict(zip(sorted(ref), np.load(tmp_path / "info.npy")))
    assert got["count"] == ref["count"] == total and got["step"] == ref["step"] == 198
    for key in ref:
        assert got[key] == pytest.approx(ref[key], rel=1e-5, abs=1e-6, nan_ok=True), key
