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


def _worker(rank, world, port, total, steps, out_dir, variant="KAMMA_medium"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle.binding import OracleEnv
    lo, hi = shard_range(total, rank, world)
    cfg = build_config(variant=variant, num_envs=hi - lo, seed=11, arena_offset=lo, max_sim_steps=80.0)
    env = OracleEnv(cfg)
    env.reset(11)
    env.rollout(steps, _abi.POLICY_RANDOM)
    parts = gather_stats(env.stats(reset_after=True))
    assert len(parts) == world
    merged = merge_stats(parts)
    np.save(os.path.join(out_dir, f"state_{rank}.npy"), env.buffer(_abi.BUF_POSE))
    info = gather_episode_info(env.episode_info(), hi - lo, eval_keys=bool(cfg.eval_sums), single_gamma=cfg.mix_mode == _abi.MIX_BLEND)
    if rank == 0:
        with open(os.path.join(out_dir, "info_keys.txt"), "w") as f:
            f.write("\n".join(info))                                  # the ORDER the merged dict has
        np.save(os.path.join(out_dir, "info.npy"), np.array([info[k] for k in sorted(info)]))
        np.save(os.path.join(out_dir, "merged.npy"), np.array([merged["episodes"], merged["return_mean"], merged["return_min"],
                                                                 merged["return_max"], merged["steps"], merged["length_mean"]]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("variant", ["KAMMA_medium", "KA_DDPG_2D_eval"])
def test_two_rank_sharded_run_matches_single_process(tmp_path, variant):
    """two gloo ranks, each with its shard of the arenas, against the unsharded run: every arena's state, the merged episode
    statistics, and the merged extras["episode"] — under the key set and in the order of the variant's own env file (KA-DDPG:
    one "gamma assist"; the eval files: three more sums), i.e. what a single-rank infos["episode"] iterates"""
    total, steps, world = 100, 110, 2
    mp.spawn(_worker, args=(world, _free_port(), total, steps, str(tmp_path), variant), nprocs=world, join=True)
    from oracle.binding import OracleEnv
    fcfg = build_config(variant=variant, num_envs=total, seed=11, max_sim_steps=80.0)
    full = OracleEnv(fcfg)
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
    layout = _abi.episode_info_layout(bool(fcfg.eval_sums), fcfg.mix_mode == _abi.MIX_BLEND)
    ref = {key: float(ei[slot]) for key, slot in layout}   # the key set a single-rank infos["episode"] has
    assert open(tmp_path / "info_keys.txt").read().split("\n") == [k for k, _ in layout] + ["count", "step"]
    assert ("gamma assist" in ref) == (variant != "KAMMA_medium") and ("rew_Smoothness" in ref) == (variant != "KAMMA_medium")
    ref["count"], ref["step"] = float(ei[_abi.EI_COUNT]), float(ei[_abi.EI_STEP])
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
    # the KA-DDPG files log ONE key "gamma assist" (KA_DDPG_4D.py:484): a sharded run's merged dict has the key set
    # and ORDER of the single-rank infos["episode"] of its variant, which is the reference's dict order (KAMMA.py:204-221, 480-488)
    ka = gather_episode_info(vec, 16, eval_keys=False, single_gamma=True)
    assert ka["gamma assist"] == 0.75 and "gamma_1 action assist" not in ka and "gamma_2 reward assist" not in ka
    assert list(ka)[:-2] == [k for k, _ in _abi.episode_info_layout(False, True)] and list(ka)[-2:] == ["count", "step"]
    order = [k for k, _ in _abi.episode_info_layout(True, False)]
    assert order[:3] == ["rew_r1: Tracking error reward (squared errors)", "rew_r2: Convergence reward (squared accelerations)",
                         "rew_r3: Torque penalty (sum of squared torques)"]
    assert order[11:15] == list(_abi.EPISODE_INFO_ZERO_KEYS) and order[15:18] == ["rew_Smoothness", "rew_Tracking error", "rew_Desired velocity"]
    assert order[18:22] == ["rew_Dense/Final reward", "rew_Sparse/Final reward", "rew_Guiding/Final reward", "rew_Observed/Final reward"]
    assert order[22:] == ["gamma_1 action assist", "gamma_2 reward assist", "terrain_level"]
    assert [k for k, _ in _abi.episode_info_layout(False, True)][-2:] == ["gamma assist", "terrain_level"]


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
    ref = {key: float(ei[slot]) for key, slot in _abi.episode_info_layout(True, False)}   # the key set a single-rank infos["episode"] has
    ref["count"], ref["step"] = float(ei[_abi.EI_COUNT]), float(ei[_abi.EI_STEP])
    got = dict(zip(sorted(ref), np.load(tmp_path / "info.npy")))
    assert got["count"] == ref["count"] == total and got["step"] == ref["step"] == 198
    for key in ref:
        assert got[key] == pytest.approx(ref[key], rel=1e-5, abs=1e-6, nan_ok=True), key


# ---------------------------------------------------------------------------------------------------------------------
# The quantities the reference defines over `num_envs` are per ENVIRONMENT, i.e. per shard: the eval variants'
# old_torques zeroing (KAMMA_eval.py:470) and the `fraction` snapshot of the level curriculum (KAMMA.py:498). A sharded
# run of those variants is the run of one reference process per shard — NOT the unsharded run. Pinned here; the env-0
# reward guard of KAMMA_g1d_g21 goes by GLOBAL arena id and therefore IS shard-invariant.
def _run_shards(make, variant, total, bounds, steps, seed, falls, **kw):
    """one environment per [lo, hi) of `bounds`, stepped in lock-step on the same global actions; `falls` = {step: global
    arena} robots put on their back before that step. Returns the concatenated per-step buffers."""
    envs = [make(build_config(variant=variant, num_envs=hi - lo, seed=seed, arena_offset=lo, **kw)) for lo, hi in bounds]
    for e in envs:
        e.reset(seed)
    rng = np.random.default_rng(seed)
    A = envs[0].cfg.action_dim
    out = []
    for t in range(steps):
        a = rng.uniform(-1, 1, (total, A)).astype(np.float32)
        for e, (lo, hi) in zip(envs, bounds):
            if t in falls and lo <= falls[t] < hi:
                e.set_quat(falls[t] - lo, [0.0, 1.0, 0.0, 0.0])
            e.step(a[lo:hi])
        out.append({k: np.concatenate([e.get(bid) for e in envs]) for k, bid in
                    (("sums_c", _abi.BUF_SUMS_C), ("rew", _abi.BUF_REW), ("done", _abi.BUF_DONE), ("commands", _abi.BUF_COMMANDS),
                     ("fraction", _abi.BUF_CMD_FRACTION), ("pose", _abi.BUF_POSE))})
        out[-1]["diag_rew"] = np.array([e.get(_abi.BUF_ENV0_DIAG)[27] for e in envs])   # the divisor of each environment's env-0 percentages
    return out


class _OrcShard:
    def __init__(self, cfg):
        from oracle.binding import OracleEnv
        self.cfg, self.e = cfg, OracleEnv(cfg)

    def reset(self, seed): self.e.reset(seed)
    def step(self, a): self.e.step(a)
    def set_quat(self, i, q): self.e.buffer(_abi.BUF_QUAT)[i] = np.array(q, np.float32)
    def get(self, bid): return np.array(self.e.buffer(bid), copy=True)


class _DevShard:
    def __init__(self, cfg):
        from options_in_simulation_amd.vecenv import NativeEnv
        self.cfg, self.e = cfg, NativeEnv(cfg, "cuda:0")

    def reset(self, seed): self.e.reset(seed)
    def step(self, a): self.e.step(torch.from_numpy(a).cuda())
    def set_quat(self, i, q): self.e.buffer(_abi.BUF_QUAT)[i] = torch.tensor(q, device="cuda")
    def get(self, bid): return self.e.buffer(bid).cpu().numpy()


def _check_eval_shard_semantics(make):
    total, steps, falls = 12, 40, {20: 2}          # arena 2 (shard 0) falls alone at step 20
    whole = _run_shards(make, "KAMMA_eval", total, [(0, total)], steps, 5, falls)
    split = _run_shards(make, "KAMMA_eval", total, [(0, 6), (6, total)], steps, 5, falls)
    for t in range(steps):
        assert np.array_equal(whole[t]["pose"], split[t]["pose"]) and np.array_equal(whole[t]["done"], split[t]["done"])  # the arenas agree
        if t <= 20:
            assert np.array_equal(whole[t]["sums_c"], split[t]["sums_c"]), t
    assert whole[20]["done"].sum() == 1 and whole[20]["done"][2] == 1
    # step 21: the reset_idx call of step 20 zeroed old_torques of every arena of ITS environment — all 12 in the
    # unsharded run, the 6 of shard 0 in the sharded one; shard 1 keeps the previous torques in its Smoothness term
    d = whole[21]["sums_c"][:, 0] != split[21]["sums_c"][:, 0]
    assert not d[:6].any() and d[6:].all()
    return split


def test_eval_variant_shards_are_per_environment_on_the_oracle():
    _check_eval_shard_semantics(_OrcShard)


def test_level_curriculum_fraction_is_per_environment_on_the_oracle():
    total, steps, falls = 8, 30, {12: 1}
    kw = dict(cmd_sampler=_abi.CMD_LEVEL)
    whole = _run_shards(_OrcShard, "KAMMA_medium", total, [(0, total)], steps, 9, falls, **kw)
    split = _run_shards(_OrcShard, "KAMMA_medium", total, [(0, 4), (4, total)], steps, 9, falls, **kw)
    assert np.array_equal(whole[11]["fraction"], split[11]["fraction"])
    # the fall in shard 0 re-snapshots `fraction` (KAMMA.py:498) of all 8 arenas in the unsharded run, of shard 0's 4 otherwise
    d = whole[12]["fraction"] != split[12]["fraction"]
    assert not d[:4].any() and d[4:].all() and (split[12]["fraction"][4:] == split[11]["fraction"][4:]).all()


def test_env0_reward_guard_goes_by_global_arena_id_on_the_oracle():
    total, steps = 8, 60   # the guard needs rew == 0: gamma2 = 1 (constant in this variant) and an unclipped expert torque executed
    whole = _run_shards(_OrcShard, "KAMMA_g1d_g21", total, [(0, total)], steps, 3, {})
    split = _run_shards(_OrcShard, "KAMMA_g1d_g21", total, [(0, 4), (4, total)], steps, 3, {})
    guarded = [t for t in range(steps) if whole[t]["rew"][0] == np.float32(1e-12)]
    assert guarded, "the guard never fired: lengthen the run"
    for t in range(steps):
        assert np.array_equal(whole[t]["rew"], split[t]["rew"]), t          # shard-invariant
        assert not (split[t]["rew"][4] == np.float32(1e-12))                # local arena 0 of shard 1 is global arena 4: not guarded
    # ... while the env-0 DIAGNOSTICS are per environment, and so is the guard of their divisor: the arena that logs on
    # shard 1 is global arena 4, whose reward is exactly 0 on some steps; observed_components must not divide by it
    zero4 = [t for t in range(steps) if whole[t]["rew"][4] == 0.0]
    assert zero4, "arena 4's reward is never exactly 0: lengthen the run"
    for t in zero4:
        assert split[t]["diag_rew"][1] == np.float32(1e-12) and split[t]["rew"][4] == 0.0
    from options_in_simulation_amd.vecenv import observed_components_from_diag
    d = np.zeros(32, np.float32); d[27], d[19], d[20] = split[zero4[0]]["diag_rew"][1], 1.0, -0.5
    assert all(np.isfinite(v) for v in observed_components_from_diag(d).values())


@pytest.mark.gpu
def test_eval_and_guard_shard_semantics_on_the_hip_path():
    """the product's shards behave as the oracle's with the same boundaries (bit for bit), incl. the per-environment
    old_torques zeroing and the global-id reward guard"""
    dev = _check_eval_shard_semantics(_DevShard)
    orc = _run_shards(_OrcShard, "KAMMA_eval", 12, [(0, 6), (6, 12)], 40, 5, {20: 2})
    for t in range(40):
        for k in ("sums_c", "rew", "done", "commands", "pose"):
            assert np.array_equal(dev[t][k], orc[t][k]), (t, k)
    a = _run_shards(_DevShard, "KAMMA_g1d_g21", 8, [(0, 4), (4, 8)], 60, 3, {})
    b = _run_shards(_OrcShard, "KAMMA_g1d_g21", 8, [(0, 8)], 60, 3, {})
    for t in range(60):
        assert np.array_equal(a[t]["rew"], b[t]["rew"]), t
