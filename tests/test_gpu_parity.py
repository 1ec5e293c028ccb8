"""HIP kernel vs CPU oracle through the C ABI, same seeded inputs (-m gpu).

Tolerance: BASELINE.json north_star asks 1e-5 relative for floats and bit-exact masks /
counters. The kernel and the oracle spell the same float32 operation order (explicit fma,
-ffp-contract=off), so most checks here are in fact exact; where a test says rtol it is the
north_star's 1e-5.
"""
import os
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

from options_in_simulation_amd import _abi, build_config  # noqa: E402



@pytest.fixture(autouse=True, params=["by-size", "one-lane"])
def lanes_per_arena(request, monkeypatch):
    """The library picks the lane layout by arena count (a lane pair per arena up to 32 768 arenas, one lane above,
    include/ois.h): every test of this file runs under that choice AND with one lane per arena forced, so that
    the headline layout keeps the coverage these small-N tests gave it."""
    if request.param == "one-lane":
        monkeypatch.setenv("OIS_LANES_PER_ARENA", "1")
    return request.param


STATE_BUFS = [_abi.BUF_POSE, _abi.BUF_QUAT, _abi.BUF_LINVEL, _abi.BUF_ANGVEL, _abi.BUF_WHEEL_VEL,
              _abi.BUF_COMMANDS, _abi.BUF_SUMS_A, _abi.BUF_SUMS_B, _abi.BUF_SUMS_C, _abi.BUF_LAMBDA_N, _abi.BUF_LAMBDA_L,
              _abi.BUF_LAMBDA_T, _abi.BUF_WHEEL_TORQS, _abi.BUF_OBS, _abi.BUF_REW]
EXACT_BUFS = [_abi.BUF_COUNTERS, _abi.BUF_DONE, _abi.BUF_TIMEOUT]
RTOL, ATOL = 1e-5, 1e-6


def make_pair(device="cuda:0", oracle_threads=1, **kw):
    from options_in_simulation_amd.vecenv import NativeEnv
    from oracle.binding import OracleEnv
    cfg = build_config(**kw)
    return cfg, NativeEnv(cfg, device), OracleEnv(cfg, threads=oracle_threads)


def compare(dev, orc, where="", exact_float=False):
    torch.cuda.synchronize()
    for bid in EXACT_BUFS:
        got = dev.buffer(bid).cpu().numpy()
        if got.dtype == np.bool_:
            got = got.astype(np.uint8)
        assert np.array_equal(got, orc.buffer(bid)), f"{where}: buffer {bid} differs (must be bit-exact)"
    worst = 0.0
    for bid in STATE_BUFS:
        got, ref = dev.buffer(bid).cpu().numpy(), orc.buffer(bid)
        if exact_float:
            assert np.array_equal(got, ref, equal_nan=True), f"{where}: buffer {bid} not bit-identical"
        np.testing.assert_allclose(got, ref, rtol=RTOL, atol=ATOL, err_msg=f"{where}: buffer {bid}")
        worst = max(worst, float(np.max(np.abs(got - ref))) if got.size else 0.0)
    return worst


@pytest.mark.parametrize("variant", ["KAMMA_medium", "KA_DDPG_4D", "KA_DDPG_2D", "KA_DDPG_1D", "KAMMA_eval",
                                     "KAMMA_curr1", "KAMMA_curr2", "KAMMA_g1d_g21"])
def test_step_parity_variants(variant):
    N = 200  # not a multiple of 16: ragged last wave
    cfg, dev, orc = make_pair(variant=variant, num_envs=N, seed=11, max_sim_steps=60.0)
    dev.reset(11); orc.reset(11)
    compare(dev, orc, f"{variant} after reset", exact_float=True)
    rng = np.random.default_rng(0)
    lo = -1.0 if variant.startswith("KA_DDPG") else 0.0
    for t in range(12):
        a = rng.uniform(lo, 1.0, size=(N, cfg.action_dim)).astype(np.float32)
        dev.step(torch.from_numpy(a).cuda()); orc.step(a)
        compare(dev, orc, f"{variant} step {t}", exact_float=True)


@pytest.mark.parametrize("variant,N", [("KAMMA_eval", 300), ("KA_DDPG_2D_eval", 300), ("KAMMA_eval", 140000)],
                         ids=["KAMMA_eval", "KA_DDPG_2D_eval", "KAMMA_eval-140000"])
def test_eval_variants_zero_old_torques_of_every_arena_on_any_reset(variant, N):
    """KAMMA_eval.py:470: reset_idx sets old_torques = zeros(num_envs, 4), so ONE robot falling changes the next
    step's Smoothness of every arena. The step kernel learns it through a stamp the resetting wave leaves for the
    next launch (two slots by step parity; a rollout of these variants is k single-step launches). Falls of single
    arenas, two in consecutive steps, the synchronous timeout, a fused rollout; at 140 000 arenas the waves of a
    launch run in several rounds (the stamp must not be read from the launch that writes it). Bit for bit."""
    cfg, dev, orc = make_pair(variant=variant, num_envs=N, seed=21)
    dev.reset(21); orc.reset(21)
    rng = np.random.default_rng(2)
    steps = 103 if N < 1000 else 9
    falls = {4: 7, 5: N - 1, 40: 123} if N < 1000 else {3: N // 2, 4: 5}
    up = np.array([0.0, 1.0, 0.0, 0.0], np.float32)
    for t in range(steps):
        if t in falls:
            orc.buffer(_abi.BUF_QUAT)[falls[t]] = up
            dev.buffer(_abi.BUF_QUAT)[falls[t]] = torch.from_numpy(up).cuda()
        a = rng.uniform(-1, 1, size=(N, cfg.action_dim)).astype(np.float32)
        dev.step(torch.from_numpy(a).cuda()); orc.step(a)
        if t in falls or t - 1 in falls or t >= 97 or t == steps - 1:
            compare(dev, orc, f"{variant} step {t}", exact_float=True)
            if orc.done.any():
                np.testing.assert_array_equal(dev.episode_info().cpu().numpy(), orc.episode_info())
    acts = rng.uniform(-1, 1, size=(6, N, cfg.action_dim)).astype(np.float32)
    orc.buffer(_abi.BUF_QUAT)[11] = up
    dev.buffer(_abi.BUF_QUAT)[11] = torch.from_numpy(up).cuda()
    rs = torch.zeros(N, device="cuda")
    dev.rollout(6, _abi.POLICY_EXTERNAL, torch.from_numpy(acts).cuda(), reward_sum=rs)
    ro = orc.rollout(6, _abi.POLICY_EXTERNAL, acts, want_reward_sum=True)
    compare(dev, orc, f"{variant} fused rollout with a fall in its first step", exact_float=True)
    assert np.array_equal(rs.cpu().numpy(), ro)


@pytest.mark.parametrize("N", [1, 1000, 66000, 140000, 524288])
def test_one_launch_logging_reductions_against_the_oracle(N):
    """extras["episode"] and the finished-episode statistics since round 4 (logs_block_kernel: one workgroup; logs_level1 / level2
    and stats_level1 / level2: two launches). Same two-level order as the oracle, bit for bit: 1 chunk, 1 ragged chunk,
    65 chunks (one workgroup, 16 waves x 2 chunks x 3 passes), 137 and 512 chunks (one wave per chunk, then three / eight level-2 batches); after post_reset (every group carries stamp 0),
    after single falls (one group, one chunk), after two falls in different chunks of one step, and with short episodes after
    the synchronous timeout. The block ois_logs_fetch publishes in pinned host memory == the device buffers == the oracle."""
    cfg, dev, orc = make_pair(variant="KAMMA_medium", num_envs=N, seed=31, max_episode_length=6,  # timeout when episode_buf reaches 5
                              oracle_threads=8 if N > 100000 else 1)
    dev.reset(31); orc.reset(31)
    rng = np.random.default_rng(8)
    up = np.array([0.0, 1.0, 0.0, 0.0], np.float32)
    falls = {1: [N // 2], 2: [0, N - 1], 3: [min(N - 1, 1024 * 64 + 5)]}

    def check(where):
        ei, diag = dev.host_logs()[:2]
        np.testing.assert_array_equal(ei, orc.episode_info(), err_msg=where)
        np.testing.assert_array_equal(ei, dev.buffer(_abi.BUF_EPISODE_INFO).cpu().numpy(), err_msg=where)
        np.testing.assert_array_equal(diag, dev.buffer(_abi.BUF_ENV0_DIAG).cpu().numpy(), err_msg=where)
        np.testing.assert_array_equal(diag, orc.buffer(_abi.BUF_ENV0_DIAG), err_msg=where)
        np.testing.assert_array_equal(dev.episode_info().cpu().numpy(), ei, err_msg=where)  # ois_episode_info: the same kernel

    check("post_reset")
    assert dev.host_logs()[0][_abi.EI_COUNT] == N
    for t in range(9):
        for i in falls.get(t, ()):
            orc.buffer(_abi.BUF_QUAT)[i] = up
            dev.buffer(_abi.BUF_QUAT)[i] = torch.from_numpy(up).cuda()
        a = rng.uniform(0, 1, size=(N, 4)).astype(np.float32)
        dev.step(torch.from_numpy(a).cuda()); orc.step(a)
        check(f"step {t}")
        if t in falls:
            assert dev.host_logs()[0][_abi.EI_COUNT] == len(set(falls[t]))
        if t == 4 and N >= 1000:  # the synchronous timeout of everyone who did not fall on the way: every chunk carries the newest stamp
            fallen = {i for ids in falls.values() for i in ids}
            assert dev.host_logs()[0][_abi.EI_COUNT] == N - len(fallen) and dev.host_logs()[0][_abi.EI_STEP] == 5
    for reset_after in (0, 1, 0):
        sd, so = dev.stats(bool(reset_after)), orc.stats(bool(reset_after))
        for f in ("episodes", "return_sum", "return_sqsum", "return_min", "return_max", "length_sum", "steps"):
            assert getattr(sd, f) == getattr(so, f), (f, reset_after)
    assert sd.episodes == 0


def test_long_rollout_with_resets_and_extras():
    """250 steps: two timeouts (99-step episodes), extras['episode'] and episode statistics."""
    N = 333
    cfg, dev, orc = make_pair(variant="KAMMA_medium", num_envs=N, seed=3, max_sim_steps=300.0)
    dev.reset(3); orc.reset(3)
    rng = np.random.default_rng(5)
    for t in range(250):
        a = rng.uniform(0, 1, size=(N, 4)).astype(np.float32)
        dev.step(torch.from_numpy(a).cuda()); orc.step(a)
        if t % 50 == 49 or t in (98, 99, 197, 198):
            compare(dev, orc, f"step {t}", exact_float=True)
            ei = dev.episode_info().cpu().numpy()
            np.testing.assert_allclose(ei, orc.episode_info(), rtol=RTOL, atol=ATOL, equal_nan=True)
    sd, so = dev.stats(), orc.stats()
    for f in ("episodes", "return_sum", "return_sqsum", "return_min", "return_max", "length_sum", "steps"):
        assert getattr(sd, f) == pytest.approx(getattr(so, f), rel=RTOL), f
    assert sd.episodes == 2 * N


def test_fused_rollout_equals_single_steps_and_oracle():
    N, K = 512, 40
    cfg, dev, orc = make_pair(variant="KAMMA_eval", num_envs=N, seed=9)
    dev.reset(9); orc.reset(9)
    rs = torch.zeros(N, device="cuda")
    dev.rollout(K, _abi.POLICY_RANDOM, reward_sum=rs)
    ro = orc.rollout(K, _abi.POLICY_RANDOM, want_reward_sum=True)
    compare(dev, orc, "random-policy rollout", exact_float=True)
    np.testing.assert_allclose(rs.cpu().numpy(), ro, rtol=RTOL, atol=ATOL)
    acts = torch.rand((K, N, 4), generator=torch.Generator().manual_seed(1))
    dev.rollout(K, _abi.POLICY_EXTERNAL, actions=acts.cuda())
    for k in range(K):
        orc.step(acts[k].numpy())
    compare(dev, orc, "external-policy rollout vs k single steps", exact_float=True)
    dev.rollout(3, _abi.POLICY_ZERO); orc.rollout(3, _abi.POLICY_ZERO)
    compare(dev, orc, "zero-policy rollout", exact_float=True)


def test_domain_randomisation_parity():
    N = 1000
    cfg, dev, orc = make_pair(variant="KAMMA_g11_g21", num_envs=N, seed=21)
    slopes = [-0.3, -0.2, 0.0, 0.2, 0.3]  # origin_terrain_generator.py:92
    dev.randomize_params(0.5, 1.0, slopes); orc.randomize_params(0.5, 1.0, slopes)
    torch.cuda.synchronize()
    assert np.array_equal(dev.buffer(_abi.BUF_PARAMS).cpu().numpy(), orc.buffer(_abi.BUF_PARAMS))
    dev.reset(21); orc.reset(21)
    z = np.zeros((N, 4), np.float32)
    for t in range(30):
        dev.step(torch.from_numpy(z).cuda()); orc.step(z)
    compare(dev, orc, "sloped arenas, expert controller", exact_float=True)


@pytest.mark.parametrize("N,shell", [(32768, "step_kernel<2,"), (32769, "step_kernel<1,"), (65536, "step_kernel<1,"), (65537, "step_kernel_w2<")])
def test_shell_selection_boundaries(N, shell, lanes_per_arena):
    """The library picks the step kernel by arena count: a lane pair per arena up to 32 768 (= one wave per SIMD of an
    MI355X), one lane per arena with one wave per SIMD claimed up to 65 536, two waves per SIMD allowed above. At each
    boundary and one arena beyond it (a last wave with a single live lane): the shell ois_kernel_info names, and three steps +
    a fused rollout bit-identical to the oracle."""
    cfg, dev, orc = make_pair(variant="KAMMA_medium", num_envs=N, seed=23, oracle_threads=8)
    if lanes_per_arena == "by-size":
        assert dev.kernel_info().startswith(shell), dev.kernel_info()
    dev.reset(23); orc.reset(23)
    rng = np.random.default_rng(N)
    for t in range(3):
        a = rng.uniform(0, 1, size=(N, 4)).astype(np.float32)
        dev.step(torch.from_numpy(a).cuda()); orc.step(a)
    compare(dev, orc, f"N={N} single steps", exact_float=True)
    dev.rollout(4, _abi.POLICY_RANDOM); orc.rollout(4, _abi.POLICY_RANDOM)
    compare(dev, orc, f"N={N} fused rollout", exact_float=True)
    np.testing.assert_array_equal(dev.host_logs()[0], orc.episode_info())
    dev.close()


def test_sharding_invariance():
    """arena i gets the same stream whether it lives in a 1-GPU env or in shard 1 of 2."""
    from options_in_simulation_amd.vecenv import NativeEnv
    N = 256
    full = NativeEnv(build_config(variant="KAMMA_medium", num_envs=N, seed=5), "cuda:0")
    half = NativeEnv(build_config(variant="KAMMA_medium", num_envs=N // 2, seed=5, arena_offset=N // 2), "cuda:0")
    full.reset(5); half.reset(5)
    a = torch.rand((N, 4), device="cuda")
    for _ in range(5):
        full.step(a); half.step(a[N // 2:])
    torch.cuda.synchronize()
    for bid in STATE_BUFS + EXACT_BUFS:
        assert torch.equal(full.buffer(bid)[N // 2:] if full.buffer(bid).shape[0] == N else full.buffer(bid),
                           half.buffer(bid) if half.buffer(bid).shape[0] == N // 2 else full.buffer(bid))


def test_caller_written_plane_normals_after_params_dirty():
    """OIS_BUF_PARAMS is writable, but the flat-plane kernels take n = (0, 0, 1) without reading it. A caller that
    does its own domain randomisation by writing normals says so (ois_params_dirty -> the general kernels); friction alone
    (column 0) needs no call. Kernel == oracle, which always honours the written rows."""
    N = 300
    cfg, dev, orc = make_pair(variant="KAMMA_medium", num_envs=N, seed=17)
    rng = np.random.default_rng(6)
    prm = np.zeros((N, 4), np.float32)
    prm[:, 0] = rng.uniform(0.5, 1.0, N)
    prm[:, 3] = 1.0
    dev.buffer(_abi.BUF_PARAMS).copy_(torch.from_numpy(prm)); orc.buffer(_abi.BUF_PARAMS)[:] = prm
    assert "plane=flat" in dev.kernel_info()          # friction only: still the flat shells
    dev.reset(17); orc.reset(17)
    a = rng.uniform(0, 1, (N, 4)).astype(np.float32)
    for t in range(4):
        dev.step(torch.from_numpy(a).cuda()); orc.step(a)
    compare(dev, orc, "own friction", exact_float=True)
    sl = rng.choice([-0.3, 0.0, 0.2], N).astype(np.float32)
    th = rng.uniform(0, np.pi, N).astype(np.float32)
    inv = (1.0 / np.sqrt(sl * sl + 1.0)).astype(np.float32)
    prm[:, 1], prm[:, 2], prm[:, 3] = -(sl * np.cos(th)) * inv, -(sl * np.sin(th)) * inv, inv
    dev.buffer(_abi.BUF_PARAMS).copy_(torch.from_numpy(prm)); orc.buffer(_abi.BUF_PARAMS)[:] = prm
    dev.params_written()
    assert "plane=per-arena" in dev.kernel_info()
    for t in range(6):
        dev.step(torch.from_numpy(a).cuda()); orc.step(a)
    compare(dev, orc, "own normals", exact_float=True)
    assert not np.array_equal(orc.buffer(_abi.BUF_POSE)[:, 2], np.full(N, orc.buffer(_abi.BUF_POSE)[0, 2]))


def test_save_load_state_roundtrip():
    N = 128
    cfg, dev, orc = make_pair(variant="KAMMA_medium", num_envs=N, seed=2)
    dev.reset(2)
    a = torch.rand((N, 4), device="cuda")
    for _ in range(4):
        dev.step(a)
    blob = dev.save_state()
    for _ in range(6):
        dev.step(a)
    ref = {b: dev.buffer(b).clone() for b in STATE_BUFS + EXACT_BUFS}
    dev.load_state(blob)
    for _ in range(6):
        dev.step(a)
    torch.cuda.synchronize()
    for b, t in ref.items():
        assert torch.equal(dev.buffer(b), t)


def test_error_paths():
    from options_in_simulation_amd.vecenv import NativeEnv
    cfg = build_config(variant="KAMMA_medium", num_envs=8)
    env = NativeEnv(cfg, "cuda:0")
    with pytest.raises(RuntimeError, match="before ois_reset"):
        env.step(torch.zeros((8, 4), device="cuda"))
    env.reset(1)
    with pytest.raises(ValueError):
        env.step(torch.zeros((7, 4), device="cuda"))
    bad = build_config(variant="KAMMA_medium", num_envs=8)
    bad.action_dim = 3
    with pytest.raises(ValueError):
        NativeEnv(bad, "cuda:0")


@pytest.mark.parametrize("env_vars", [{"OIS_PACKED": "0", "OIS_LANES_PER_ARENA": "1"}, {"OIS_PACKED": "1", "OIS_LANES_PER_ARENA": "1"},
                                      {"OIS_LANES_PER_ARENA": "2"}, {"OIS_LANES_PER_ARENA": "2", "OIS_PACKED": "0"},
                                      {"OIS_PACKED": "0", "OIS_LANES_PER_ARENA": "1", "OIS_STEP_WAVES": "2"},
                                      {"OIS_PACKED": "1", "OIS_LANES_PER_ARENA": "1", "OIS_STEP_WAVES": "2"}],
                         ids=["scalar", "packed", "lane-pair-packed", "lane-pair-scalar", "scalar-256reg", "packed-256reg"])
def test_every_kernel_instantiation_is_bit_identical_to_the_oracle(env_vars, monkeypatch):
    """step_kernel<1,false>, <1,true>, <2,true>, <2,false> and the 256-register shells step_kernel_w2 (the
    default above 65 536 arenas) share one arithmetic (docs/SPEC.md); the library picks by size,
    the environment variables (read at ois_create) force each one."""
    for k, v in env_vars.items():
        monkeypatch.setenv(k, v)
    N = 300
    cfg, dev, orc = make_pair(variant="KAMMA_medium", num_envs=N, seed=17, max_sim_steps=150.0)
    dev.reset(17); orc.reset(17)
    rng = np.random.default_rng(3)
    for t in range(105):  # crosses the timeout at step 99
        a = rng.uniform(0, 1, size=(N, 4)).astype(np.float32)
        dev.step(torch.from_numpy(a).cuda()); orc.step(a)
        if t % 20 == 0 or t >= 98:
            compare(dev, orc, f"{env_vars} step {t}", exact_float=True)
    np.testing.assert_array_equal(dev.episode_info().cpu().numpy(), orc.episode_info())
    dev.rollout(7, _abi.POLICY_RANDOM); orc.rollout(7, _abi.POLICY_RANDOM)
    compare(dev, orc, f"{env_vars} fused rollout", exact_float=True)


@pytest.mark.parametrize("env_vars", [{"OIS_PACKED": "0", "OIS_LANES_PER_ARENA": "1"}, {"OIS_PACKED": "1", "OIS_LANES_PER_ARENA": "1"},
                                      {"OIS_LANES_PER_ARENA": "2"}, {"OIS_LANES_PER_ARENA": "2", "OIS_PACKED": "0"}],
                         ids=["scalar", "packed", "lane-pair-packed", "lane-pair-scalar"])
def test_non_finite_outcome_guard_parity(env_vars, monkeypatch):
    """docs/SPEC.md §3.9: injected NaN / inf velocities, and a finite linear speed beyond 50 m/s (the launch a Newton refresh in
    a singular pose can produce) -> the arena is reset as fallen, outputs
    finite, kernel == oracle bit for bit (also for the untouched arenas). Lane-pair layout on the
    plane only; the other two also on the height field (NaN positions must not index outside it)."""
    for k, v in env_vars.items():
        monkeypatch.setenv(k, v)
    for field in ((False,) if env_vars.get("OIS_LANES_PER_ARENA") == "2" else (False, True)):
        _guard_case(env_vars, field)


def _guard_case(env_vars, field):
    from options_in_simulation_amd import terrain as tg
    N = 130
    cfg, dev, orc = make_pair(variant="KAMMA_medium", num_envs=N, seed=4)
    if field:
        T = tg.Terrain({"numLevels": 2, "numTerrains": 2, "mapLength": 8.0, "mapWidth": 8.0}, N, seed=5)
        lt = T.sample_levels_types(N, rng=np.random.RandomState(6))
        dev.attach_terrain(T, lt); orc.attach_terrain(T, lt)
    dev.reset(4); orc.reset(4)
    z = np.zeros((N, 4), np.float32)
    for _ in range(2):
        dev.step(torch.from_numpy(z).cuda()); orc.step(z)
    torch.cuda.synchronize()
    for bid, i, j, val in [(_abi.BUF_LINVEL, 3, 1, np.nan), (_abi.BUF_ANGVEL, 64, 0, np.inf), (_abi.BUF_WHEEL_VEL, 129, 2, -np.inf),
                           (_abi.BUF_QUAT, 77, 3, np.nan), (_abi.BUF_POSE, 100, 0, np.nan), (_abi.BUF_POSE, 101, 1, np.inf),
                           (_abi.BUF_LINVEL, 10, 2, 80.0), (_abi.BUF_LINVEL, 11, 2, 30.0)]:   # straight up: 80 m/s is discarded, 30 m/s is a flight
        dev.buffer(bid)[i, j] = float(val); orc.buffer(bid)[i, j] = val
    dev.step(torch.from_numpy(z).cuda()); orc.step(z)
    worst = compare(dev, orc, f"{env_vars} guard step", exact_float=True)
    assert np.isfinite(worst)
    done = dev.buffer(_abi.BUF_DONE).cpu().numpy().astype(bool)
    assert done[[3, 64, 129, 77, 100, 101, 10]].all() and done.sum() == 7 and not done[11]
    for bid in STATE_BUFS:
        assert torch.isfinite(dev.buffer(bid)).all(), bid
    for _ in range(3):
        dev.step(torch.from_numpy(z).cuda()); orc.step(z)
    compare(dev, orc, f"{env_vars} after the guard", exact_float=True)


@pytest.mark.parametrize("env_vars", [{"OIS_PACKED": "0", "OIS_LANES_PER_ARENA": "1"}, {"OIS_PACKED": "1", "OIS_LANES_PER_ARENA": "1"},
                                      {"OIS_PACKED": "1", "OIS_LANES_PER_ARENA": "1", "OIS_STEP_WAVES": "2"}, {"OIS_LANES_PER_ARENA": "2"}],
                         ids=["scalar", "packed", "packed-256reg", "lane-pair-packed"])
@pytest.mark.parametrize("terrain", [False, True], ids=["plane", "field"])
def test_runtime_iteration_count_instantiations(env_vars, terrain, monkeypatch):
    """solver_iterations != 4 selects the NIT = 0 kernels (run-time loop instead of the unrolled
    four iterations): same arithmetic, checked against the oracle like the default ones."""
    from options_in_simulation_amd import terrain as tg
    if terrain and env_vars.get("OIS_LANES_PER_ARENA") == "2":
        pytest.skip("the terrain contact needs all four wheels in one lane")
    for k, v in env_vars.items():
        monkeypatch.setenv(k, v)
    N = 140
    cfg, dev, orc = make_pair(variant="KAMMA_medium", num_envs=N, seed=21, max_sim_steps=80.0)
    assert cfg.solver_iterations == 4
    cfg.solver_iterations = 3
    from options_in_simulation_amd.vecenv import NativeEnv
    from oracle.binding import OracleEnv
    dev, orc = NativeEnv(cfg, "cuda:0"), OracleEnv(cfg)
    if terrain:
        T = tg.Terrain({"numLevels": 2, "numTerrains": 2, "mapLength": 8.0, "mapWidth": 8.0}, N, seed=9)
        lt = T.sample_levels_types(N, rng=np.random.RandomState(2))
        dev.attach_terrain(T, lt); orc.attach_terrain(T, lt)
    dev.reset(21); orc.reset(21)
    rng = np.random.default_rng(8)
    for t in range(25):
        a = rng.uniform(0, 1, size=(N, 4)).astype(np.float32)
        dev.step(torch.from_numpy(a).cuda()); orc.step(a)
    compare(dev, orc, f"{env_vars} iterations=3 terrain={terrain}", exact_float=True)
    dev.rollout(6, _abi.POLICY_RANDOM); orc.rollout(6, _abi.POLICY_RANDOM)
    compare(dev, orc, "fused", exact_float=True)


def test_second_reset_with_another_seed():
    """ois_reset in the middle of a run: everything restarts (counters, sums, statistics, impulse
    cache), the episode stream follows the new seed, parameters set by ois_randomize_params stay"""
    N = 150
    cfg, dev, orc = make_pair(variant="KAMMA_medium", num_envs=N, seed=1)
    dev.randomize_params(0.6, 0.9, [-0.2, 0.2]); orc.randomize_params(0.6, 0.9, [-0.2, 0.2])
    dev.reset(1); orc.reset(1)
    rng = np.random.default_rng(1)
    for t in range(5):
        a = rng.uniform(0, 1, size=(N, 4)).astype(np.float32)
        dev.step(torch.from_numpy(a).cuda()); orc.step(a)
    first = dev.buffer(_abi.BUF_POSE).clone()
    dev.reset(99); orc.reset(99)
    compare(dev, orc, "after the second reset", exact_float=True)
    assert not torch.equal(dev.buffer(_abi.BUF_POSE), first)
    assert (dev.buffer(_abi.BUF_COUNTERS)[:, 1] == 0).all() and (dev.buffer(_abi.BUF_COUNTERS)[:, 2] == 1).all()
    torch.cuda.synchronize()
    assert np.array_equal(dev.buffer(_abi.BUF_PARAMS).cpu().numpy(), orc.buffer(_abi.BUF_PARAMS))
    for t in range(5):
        a = rng.uniform(0, 1, size=(N, 4)).astype(np.float32)
        dev.step(torch.from_numpy(a).cuda()); orc.step(a)
    compare(dev, orc, "steps after the second reset", exact_float=True)
    sd, so = dev.stats(), orc.stats()
    assert sd.episodes == so.episodes == 0 and sd.steps == so.steps == 5 * N


def test_lean_exact_reciprocal_is_the_ieee_quotient(tmp_path):
    """exact_rcp / exact_rcp2 (ois_physics.h, ois_physics_pk.h): v_rcp_f32 + the refinement chain
    of the compiler's fdiv expansion, without its exponent-extreme handling. The oracle computes
    1.0f / k; the kernels must give the same bits: exhaustive over every float in [2^-40, 2^40)
    and the negatives (tools/exact_rcp_check.hip, scalar and packed forms)."""
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "rcpchk")
    subprocess.run([hipcc, "-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-w", "-o", exe,
                    os.path.join(root, "tools", "exact_rcp_check.hip")], check=True, capture_output=True)
    res = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0 and " 0 mismatches" in res.stdout, res.stdout + res.stderr


@pytest.mark.gpu
def test_calibration_statistic_on_the_hip_path():
    """the calibration statistic of tests/test_oracle_rng_physics.py (expert-only run against the
    reference's training log and controller figure, tests/golden/ref_stats.json) computed on the HIP
    path at 4 096 arenas: calibration, not parity — the parity of this run is the rest of this file."""
    import json
    from tests.calibration import expert_run_stats
    ref = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "ref_stats.json")))
    log, built = ref["train_log"]["first_100000_steps"], ref["build"]
    N = 4096
    from options_in_simulation_amd.vecenv import NativeEnv
    dev = NativeEnv(build_config(variant="KAMMA_g11_g21", num_envs=N, seed=11), "cuda:0")
    dev.reset(11)

    class Adapter:
        def step(self, a): dev.step(torch.from_numpy(a).cuda())
        def buffer(self, bid): return dev.buffer(bid)
    r = expert_run_stats(Adapter(), N, to_numpy=lambda t: t.cpu().numpy())
    line = lambda d, x: d["slope"] * x + d["intercept"]
    assert r["v_delta"] == pytest.approx(built["v_delta"], rel=0.05) and r["v_delta"] == pytest.approx(log["v_delta"], rel=0.05)
    assert r["slope"] == pytest.approx(built["slope"], rel=0.05)
    t = log["v_delta_vs_desired_v"]
    assert abs(line(r, 1.3) - line(t, 1.3)) < 0.02 and abs(line(r, 1.7) - line(t, 1.7)) < 0.02
    te, fig = r["te_by_command"], ref["figure_controller"]["te"]
    assert te[2] < te[0] and te[2] < te[4]
    assert abs(te[0] - fig["1.0-1.3"]) < 0.06 and abs(te[2] - fig["1.53"]) < 0.03 and abs(te[4] - fig["2.0"]) < 0.06


def test_save_load_with_replay_and_terrain_attached():
    """VERDICT r1 #8 / ADVICE: the blob carries the transition-memory cursor and says what it was saved
    under; a restored run is the run it was saved from, and a blob from another config, terrain
    attachment or memory size is refused with the reason."""
    from options_in_simulation_amd import terrain as tg
    from options_in_simulation_amd.vecenv import NativeEnv
    N, M = 96, 7
    cfg = build_config(variant="KAMMA_medium", num_envs=N, seed=8, max_sim_steps=90.0)
    T = tg.Terrain({"curriculum": True, "numLevels": 4, "numTerrains": 2, "mapLength": 8.0, "mapWidth": 8.0}, N, seed=0)
    lt = np.stack([np.arange(N) % 4, np.arange(N) % 2], 1).astype(np.int32)
    kw = dict(curriculum=True, threshold_low=-600.0, threshold_high=-330.0, max_level=3)

    def fresh(attach_terrain=True, memory=M, **cfg_kw):
        e = NativeEnv(build_config(variant="KAMMA_medium", num_envs=N, seed=8, max_sim_steps=90.0, **cfg_kw), "cuda:0")
        if attach_terrain:
            e.attach_terrain(T, lt, **kw)
        e.reset(8)
        mem = e.attach_replay(memory) if memory else None
        return e, mem
    g = torch.Generator(device="cuda").manual_seed(1)
    acts = torch.rand((130, N, 4), device="cuda", generator=g)
    a, mem_a = fresh()
    for t in range(103):  # past the timeout: levels moved, the ring wrapped many times
        a.step(acts[t])
    blob = a.save_state()
    cur = a.replay_cursor()
    assert cur == (103 % M, True)
    snap_mem = {k: v.clone() for k, v in mem_a.items()}
    for t in range(103, 130):
        a.step(acts[t])
    torch.cuda.synchronize()
    want = {b: a.buffer(b).clone() for b in STATE_BUFS + EXACT_BUFS + [_abi.BUF_TERRAIN_IDX, _abi.BUF_ENV_ORIGINS, _abi.BUF_CMD_FRACTION]}
    want_mem = {k: v.clone() for k, v in mem_a.items()}
    # a second environment, same attachments: load, restore the caller-owned memory tensors, continue
    b, mem_b = fresh()
    b.load_state(blob)
    assert b.replay_cursor() == cur
    for k in mem_b:
        mem_b[k].copy_(snap_mem[k])
    for t in range(103, 130):
        b.step(acts[t])
    torch.cuda.synchronize()
    for bid, tns in want.items():
        assert torch.equal(b.buffer(bid), tns), f"buffer {bid} after load + continue"
    for k in mem_b:
        assert torch.equal(mem_b[k], want_mem[k]), f"transition memory {k}"
    assert b.replay_cursor() == a.replay_cursor()
    # refusals, each with its reason
    for make, why in ((lambda: fresh(attach_terrain=False), "terrain attached"),
                      (lambda: fresh(memory=M + 1), "transition memory"),
                      (lambda: fresh(memory=0), "transition memory"),
                      (lambda: fresh(friction=0.6), "different ois_config"),
                      (lambda: fresh(arena_offset=N), "different ois_config")):
        c, _ = make()
        with pytest.raises(RuntimeError, match=why):
            c.load_state(blob)
        c.close()
    # the footprint stride is part of the terrain's identity: the same field under the other contact is another terrain
    c = NativeEnv(build_config(variant="KAMMA_medium", num_envs=N, seed=8, max_sim_steps=90.0), "cuda:0")
    c.attach_terrain(T, lt, footprint_stride=2, **kw)
    c.reset(8); c.attach_replay(M)
    with pytest.raises(RuntimeError, match="terrain attached"):
        c.load_state(blob)
    c.close()
    plane, _ = fresh(attach_terrain=False, memory=0)
    pb = plane.save_state()
    with pytest.raises(RuntimeError, match="saved on the plane"):
        fresh(memory=0)[0].load_state(pb)


def test_host_log_caches_follow_reset_and_load_state():
    """the host-side caches of the env-0 diagnostics / extras were keyed on policy_steps, which
    reset() sets back to 0: a second run returned the first run's values at equal step indices."""
    from options_in_simulation_amd.vecenv import VecEnv
    N = 32
    env = VecEnv(num_envs=N, device="cuda:0", variant="KAMMA_medium", seed=1)
    a = torch.full((N, 4), 0.5, device="cuda")
    env.reset(seed=1)
    env.step(a)
    d1 = dict(env.task.observed_components)
    env.reset(seed=2)  # other seed: other commands for arena 0
    env.step(a)
    d2 = dict(env.task.observed_components)
    assert d1["env0_desired_v"] != d2["env0_desired_v"]
    assert d2["env0_desired_v"] == pytest.approx(float(env.task.commands[0, 0]))
    blob = env.task.native.save_state()
    env.step(a)
    d3 = dict(env.task.observed_components)
    env.task.native.load_state(blob)
    assert dict(env.task.observed_components) == d2 != d3   # the diagnostics block is part of the state
    # extras of an earlier step cannot be read after the environment moved on
    _, _, _, _, infos = env.step(a)
    env.step(a)
    with pytest.raises(RuntimeError, match="moved on"):
        infos["episode"]["rew_Final reward"]
    _, _, _, _, infos = env.step(a)
    assert "rew_Final reward" in infos["episode"]
    env.close()


def test_every_device_ordinal_of_the_box():
    """What a multi-GPU node must not be the first to run: on EVERY ordinal torch sees (one on the test box, eight on the
    scaling node — the round-end GPUTEST then covers them all) the step parity against the oracle, the partition
    invariance of a shard living on that device against the whole on device 0, the published host logs, and the caller's
    current device left alone."""
    from options_in_simulation_amd.vecenv import NativeEnv
    ndev = torch.cuda.device_count()
    assert ndev >= 1
    N = 200
    full = NativeEnv(build_config(variant="KAMMA_medium", num_envs=N, seed=5), "cuda:0")
    full.reset(5)
    rng = np.random.default_rng(3)
    acts = rng.uniform(0, 1, size=(6, N, 4)).astype(np.float32)
    for a in acts:
        full.step(torch.from_numpy(a).to("cuda:0"))
    for d in range(ndev):
        dev_name = f"cuda:{d}"
        torch.cuda.set_device(0)
        cfg, dev, orc = make_pair(device=dev_name, variant="KAMMA_medium", num_envs=N, seed=11, max_sim_steps=60.0)
        dev.reset(11); orc.reset(11)
        for t, a in enumerate(acts):
            dev.step(torch.from_numpy(a).to(dev_name)); orc.step(a)
            compare(dev, orc, f"{dev_name} step {t}", exact_float=True)
        np.testing.assert_array_equal(dev.host_logs()[0], orc.episode_info())
        assert dev.buffer(_abi.BUF_OBS).device.index == d and torch.cuda.current_device() == 0
        half = NativeEnv(build_config(variant="KAMMA_medium", num_envs=N // 2, seed=5, arena_offset=N // 2), dev_name)
        half.reset(5)
        for a in acts:
            half.step(torch.from_numpy(a[N // 2:]).to(dev_name))
        for bid in STATE_BUFS + EXACT_BUFS:
            assert torch.equal(full.buffer(bid)[N // 2:].cpu(), half.buffer(bid).cpu()), (dev_name, bid)
        s1 = dev.stats()
        assert s1.steps == 6 * N
        dev.close(); half.close()


def test_entry_points_restore_the_callers_device():
    """a library call must not change the thread's current device (hipSetDevice guard)"""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs: with one the env's device is the caller's device anyway")
    from options_in_simulation_amd.vecenv import NativeEnv
    env = NativeEnv(build_config(variant="KAMMA_medium", num_envs=64, seed=1), "cuda:1")
    torch.cuda.set_device(0)
    env.reset(1)
    env.step(torch.rand((64, 4), device="cuda:1"))
    env.stats(); env.episode_info(); env.save_state()
    assert torch.cuda.current_device() == 0


def test_misaligned_actions_are_refused():
    from options_in_simulation_amd.vecenv import NativeEnv
    import ctypes as C
    from options_in_simulation_amd import _native
    env = NativeEnv(build_config(variant="KAMMA_medium", num_envs=64, seed=1), "cuda:0")
    env.reset(1)
    buf = torch.zeros(64 * 4 + 1, device="cuda")
    rc = env.L.ois_step(env.h, C.c_void_p(buf.data_ptr() + 4), None)
    assert rc == _abi.EINVAL and b"aligned" in env.L.ois_last_error()
    assert env.L.ois_step(env.h, C.c_void_p(buf.data_ptr()), None) == 0
    torch.cuda.synchronize()


def test_patched_task_cfg_parity_on_the_hip_path():
    """The TASK_CFG leaves of the lock-step run `KAMMA+config` (tests/golden/refexec_cosim.npz: the reference's own statements
    against the oracle) on the kernel: controller gain, action scale, 5 substeps of 0.01 s + 10, 80-step episodes, observation
    clip 5.0, a command range whose small draws are zeroed — kernel and oracle bit for bit across two timeouts."""
    from options_in_simulation_amd import default_task_cfg
    from options_in_simulation_amd.vecenv import NativeEnv
    from oracle.binding import OracleEnv
    tc = default_task_cfg()
    env = tc["task"]["env"]
    env["control"].update(stiffness=0.05, actionScale=2.0, decimation=5)
    env["learn"]["episodeLength_s"] = 4.0
    env["clipObservations"] = 5.0
    env["randomCommandVelocityRanges"]["linear_x"] = [0.0, 0.6]
    N = 1000
    cfg = build_config(tc, variant="KAMMA", num_envs=N, seed=21, max_sim_steps=300.0)
    assert cfg.max_episode_length == 80 and cfg.decimation == 5
    dev, orc = NativeEnv(cfg, "cuda:0"), OracleEnv(cfg)
    dev.reset(21); orc.reset(21)
    rng = np.random.default_rng(3)
    for t in range(165):
        a = rng.uniform(-0.2, 1.5, size=(N, 4)).astype(np.float32)
        dev.step(torch.from_numpy(a).cuda()); orc.step(a)
        if t % 20 == 0 or t in (78, 79, 80, 158, 159):
            compare(dev, orc, f"patched config step {t}", exact_float=True)
    compare(dev, orc, "patched config end", exact_float=True)
    cnt = dev.buffer(_abi.BUF_COUNTERS).cpu().numpy()
    assert (cnt[:, 2] >= 3).all()                        # post_reset + two synchronous timeouts at 79-step episodes
    cmd = dev.buffer(_abi.BUF_COMMANDS).cpu().numpy()
    assert ((cmd[:, 1] == 0) & (cmd[:, 3] == 0)).any()   # the small-command mask fired for some arenas
    assert float(dev.buffer(_abi.BUF_OBS).abs().max()) <= 5.0
