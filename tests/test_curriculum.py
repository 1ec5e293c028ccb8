"""Per-terrain-level command curriculum: sample_velocity_command's
`if self.curriculum:` branch (KAMMA.py:533-567), called for every arena after every step
(KAMMA.py:719-725) and skipped at a reset (KAMMA.py:490), with `fraction` from KAMMA.py:498 and
episode_step_counter from KAMMA.py:376, 684.

Pinned: the generators against tests/golden/refmath_curriculum.npz (a torch / Python transcription of
the cited lines in tests/golden/make_golden.py; the reference module cannot be imported here). The
RNG stream is the build's (Philox, purpose 6); torch.normal's draw enters the fixture as a given
standard normal. The reference never ran this branch with a terrain (every terrain call is commented
out, KAMMA.py:296-349): with terrain.curriculum and no terrain all arenas are level 0.
"""
import ctypes as C
import os

import numpy as np
import pytest

from options_in_simulation_amd import _abi, build_config, default_task_cfg
from options_in_simulation_amd import terrain as tg
from oracle import binding as ob
from oracle.binding import OracleEnv

G = os.path.join(os.path.dirname(__file__), "golden")


def test_level_generators_match_the_transcription():
    L = ob.lib()
    d = np.load(os.path.join(G, "refmath_curriculum.npz"))
    n = len(d["z"])
    c0, c2 = np.zeros(n, np.float32), np.zeros(n, np.float32)
    assert np.array_equal(d["fraction"], d["r1_sums"] / d["threshold_high"])  # KAMMA.py:498, float32 division
    for i in range(n):
        a, b = C.c_float(), C.c_float()
        L.orc_level_command(int(d["terrain_levels"][i]), float(d["fraction"][i]), int(d["episode_step_counter"][i]),
                            float(d["z"][i]), float(d["dt"]), float(d["max_episode_length_s"]), C.byref(a), C.byref(b))
        c0[i], c2[i] = a.value, b.value
    lv = d["terrain_levels"]
    for level in (0, 2, 3, 4):  # float32 / double arithmetic only: exact
        m = lv == level
        assert m.sum() > 20
        assert np.array_equal(c0[m], d["commands_0"][m]) and np.array_equal(c2[m], d["commands_2"][m]), level
    m = lv == 1  # math.sin in double there, a float32 polynomial here (arguments up to ~5e3 rad)
    np.testing.assert_allclose(c0[m], d["commands_0"][m], rtol=0, atol=5e-7)
    assert not c2[m].any()
    # level 0 returns (0.0, max(x_vel, 0.0)): the speed lands in commands[:, 2] (KAMMA.py:541, 724-725)
    m = lv == 0
    assert not c0[m].any() and (c2[m] > 0).mean() > 0.9


def test_deterministic_sin_over_the_argument_range():
    L = ob.lib()
    x = np.concatenate([np.linspace(-6000.0, 6000.0, 20001), np.linspace(-4.0, 4.0, 4001)]).astype(np.float32)
    got = np.array([L.orc_sin(float(v)) for v in x], np.float32)
    np.testing.assert_allclose(got, np.sin(x.astype(np.float64)), rtol=0, atol=4e-7)


def test_build_config_maps_terrain_curriculum_to_the_level_sampler():
    cfg = default_task_cfg()
    assert build_config(cfg, num_envs=2).cmd_sampler == _abi.CMD_RAND
    cfg["task"]["env"]["terrain"]["curriculum"] = True
    c = build_config(cfg, num_envs=2)
    assert c.cmd_sampler == _abi.CMD_LEVEL and c.curriculum_threshold_high == 10.0


def _level_env(N=24, seed=5, variant="KAMMA_g11_g21"):
    cfg = default_task_cfg()
    cfg["task"]["env"]["terrain"]["curriculum"] = True
    return OracleEnv(build_config(cfg, variant=variant, num_envs=N, seed=seed))


def test_commands_are_redrawn_every_step_and_not_at_a_reset():
    N = 24
    env = _level_env(N)
    env.reset(5)
    cmd = env.buffer(_abi.BUF_COMMANDS)
    frac = env.buffer(_abi.BUF_CMD_FRACTION)
    assert not frac.any()
    c_reset = cmd.copy()
    assert (c_reset[:, 0] >= 1.0).all() and not c_reset[:, 2].any()  # reset_idx draws (:448), no re-draw (:490)
    z = np.zeros((N, 4), np.float32)
    seen = []
    for t in range(1, 99):
        before = cmd.copy()
        env.step(z)
        # no terrain: every arena is level 0 -> (0.0, max(N(0.5, 0.01 + 0.09 fraction), 0)) (:536-541)
        assert not cmd[:, 0].any()
        assert np.abs(cmd[:, 2] - 0.5).max() < 0.08  # fraction stays 0 until a reset: sigma = 0.01
        assert np.array_equal(env.obs[:, 0], before[:, 0]) and np.array_equal(env.obs[:, 1], before[:, 2])  # obs precede the re-draw (:695 vs :719)
        assert not frac.any()
        seen.append(cmd[:, 2].copy())
    assert np.std(np.array(seen), axis=0).min() > 1e-3  # a new draw every step
    before = cmd.copy()
    sums_before = env.buffer(_abi.BUF_SUMS_A)[:, 0].copy()
    env.step(z)  # step 99: every arena times out
    assert env.done.all() and sums_before.min() < 0
    # the reset step's observation shows reset_idx's commands: x from :448 times the mask, yaw command = old value times the mask
    assert (env.obs[:, 0] >= 1.0).all() and np.array_equal(env.obs[:, 1], before[:, 2])
    assert not frac.any()  # :498 after the sums of the resetting arenas were zeroed (:485): all arenas reset here
    assert not cmd[:, 0].any() and (cmd[:, 2] > 0.4).all()  # and the tail re-draws right away (:719)


def test_fraction_is_resnapshot_for_all_arenas_when_any_arena_resets():
    N = 20
    env = _level_env(N, seed=9)
    env.reset(9)
    z = np.zeros((N, 4), np.float32)
    frac = env.buffer(_abi.BUF_CMD_FRACTION)
    for _ in range(30):
        env.step(z)
    assert not frac.any()
    env.buffer(_abi.BUF_QUAT)[3] = [0.0, 1.0, 0.0, 0.0]  # arena 3 on its back: has_fallen (:749-753)
    env.step(z)
    assert env.done[3] and env.done.sum() == 1
    sa = env.buffer(_abi.BUF_SUMS_A)[:, 0]
    assert np.array_equal(frac, sa / np.float32(10.0)) and frac[3] == 0.0 and (np.delete(frac, 3) < 0).all()
    snap = frac.copy()
    env.step(z)
    assert np.array_equal(frac, snap)  # nobody reset: the snapshot stays although the sums moved on
    assert not np.array_equal(snap, sa / np.float32(10.0))
    # sigma = 0.01 + 0.09 fraction is now negative for most arenas: the reference's torch.normal would
    # raise; the build keeps evaluating mean + sigma z (documented in orc_level_command)
    assert np.isfinite(env.buffer(_abi.BUF_COMMANDS)).all()


def test_levels_pick_their_generator_on_a_terrain_and_rollout_equals_steps():
    N = 48
    cfg = build_config(variant="KAMMA_medium", num_envs=N, seed=4)
    T = tg.Terrain({"curriculum": True, "numLevels": 5, "numTerrains": 2, "mapLength": 8.0, "mapWidth": 8.0}, N, seed=0)
    lt = np.stack([np.arange(N) % 5, np.arange(N) % 2], 1).astype(np.int32)
    kw = dict(curriculum=True, threshold_low=-1e9, threshold_high=1e9, max_level=4)  # levels stay
    a, b = OracleEnv(cfg), OracleEnv(cfg)
    a.attach_terrain(T, lt, **kw); b.attach_terrain(T, lt, **kw)
    a.reset(4); b.reset(4)
    rng = np.random.default_rng(0)
    acts = rng.uniform(0, 1, (12, N, 4)).astype(np.float32)
    for t in range(12):
        a.step(acts[t])
    rs = b.rollout(12, _abi.POLICY_EXTERNAL, acts, want_reward_sum=True)
    for bid in (_abi.BUF_COMMANDS, _abi.BUF_POSE, _abi.BUF_CMD_FRACTION, _abi.BUF_SUMS_B):
        assert np.array_equal(a.buffer(bid), b.buffer(bid))
    assert np.isfinite(rs).all()
    cmd, lv = a.buffer(_abi.BUF_COMMANDS), lt[:, 0]
    assert not cmd[lv == 0, 0].any() and (cmd[lv == 0, 2] > 0).all()
    assert not cmd[lv > 0, 2].any()
    t = 12 * 0.1
    np.testing.assert_allclose(cmd[lv == 1, 0], 0.5 + 0.1 * np.sin(0.01 * t), atol=1e-6)  # fraction 0: freq 0.01, amp 0.1
    np.testing.assert_allclose(cmd[lv == 2, 0], 0.5 * t / 10.0, atol=1e-6)                 # Noise = 0
    assert np.abs(cmd[lv >= 3, 0] - 0.5).max() < 0.6 and cmd[lv >= 3, 0].std() > 0.02      # N(0.5, 0.1)


# ------------------------------------------------------------------------------------- gpu
torch = pytest.importorskip("torch")

STATE = [_abi.BUF_POSE, _abi.BUF_QUAT, _abi.BUF_LINVEL, _abi.BUF_ANGVEL, _abi.BUF_WHEEL_VEL, _abi.BUF_COMMANDS,
         _abi.BUF_SUMS_A, _abi.BUF_SUMS_B, _abi.BUF_LAMBDA_N, _abi.BUF_LAMBDA_L, _abi.BUF_LAMBDA_T, _abi.BUF_WHEEL_TORQS,
         _abi.BUF_OBS, _abi.BUF_REW, _abi.BUF_ENV_ORIGINS, _abi.BUF_COUNTERS, _abi.BUF_DONE, _abi.BUF_TIMEOUT,
         _abi.BUF_TERRAIN_IDX, _abi.BUF_CMD_FRACTION]


def same_state(dev, orc, where):
    torch.cuda.synchronize()
    for bid in STATE:
        got = dev.buffer(bid).cpu().numpy()
        if got.dtype == np.bool_:
            got = got.astype(np.uint8)
        assert np.array_equal(got, orc.buffer(bid)), f"{where}: buffer {bid} not bit-identical"


@pytest.mark.gpu
@pytest.mark.parametrize("stride", [1, 2], ids=["footprints every substep", "footprints every 2nd substep"])
def test_level_curriculum_parity_on_the_field_across_level_changes(stride):
    """HIP path vs oracle, bit for bit, on the height field with curriculum = 1: the level generators,
    the fraction snapshot, update_SI_level moving arenas both ways at the timeouts, a fall in between
    (a reset of ONE arena re-snapshots every arena's fraction), single steps and a fused rollout (the FOLD shells of
    step_kernel_terrain, for both footprint strides)."""
    from options_in_simulation_amd.vecenv import NativeEnv
    N = 333
    cfg = build_config(variant="KAMMA_medium", num_envs=N, seed=7, max_sim_steps=300.0)
    dev, orc = NativeEnv(cfg, "cuda:0"), OracleEnv(cfg)
    T = tg.Terrain({"curriculum": True, "numLevels": 5, "numTerrains": 2, "mapLength": 8.0, "mapWidth": 8.0}, N, seed=0)
    lt = np.stack([np.arange(N) % 5, np.arange(N) % 2], 1).astype(np.int32)
    # r1 sums of a 99-step episode under these commands: about -3200 on level 0 (a 0.5 rad/s yaw command
    # the robot cannot follow), -940 ... -130 on the others
    kw = dict(curriculum=True, threshold_low=-600.0, threshold_high=-330.0, max_level=4, footprint_stride=stride)
    dev.attach_terrain(T, lt, **kw); orc.attach_terrain(T, lt, **kw)
    dev.reset(7); orc.reset(7)
    same_state(dev, orc, "post_reset")
    rng = np.random.default_rng(1)
    for t in range(205):
        a = rng.uniform(0, 1, (N, 4)).astype(np.float32)
        if t == 140:  # one arena falls between the timeouts
            q = np.array([0.0, 1.0, 0.0, 0.0], np.float32)
            orc.buffer(_abi.BUF_QUAT)[11] = q
            dev.buffer(_abi.BUF_QUAT)[11] = torch.from_numpy(q).cuda()
        dev.step(torch.from_numpy(a).cuda()); orc.step(a)
        if t in (0, 1, 97, 98, 99, 139, 140, 141, 196, 197, 198, 204):
            same_state(dev, orc, f"step {t}")
    moved = orc.buffer(_abi.BUF_TERRAIN_IDX)[:, 0] - lt[:, 0]
    assert (moved > 0).any() and (moved < 0).any(), "the thresholds must move arenas both ways"
    assert orc.buffer(_abi.BUF_CMD_FRACTION).any()
    np.testing.assert_array_equal(dev.episode_info().cpu().numpy(), orc.episode_info())
    acts = rng.uniform(0, 1, (9, N, 4)).astype(np.float32)
    rsum = torch.zeros(N, device="cuda:0")
    dev.rollout(9, _abi.POLICY_EXTERNAL, torch.from_numpy(acts).cuda(), reward_sum=rsum)
    ref = orc.rollout(9, _abi.POLICY_EXTERNAL, acts, want_reward_sum=True)
    same_state(dev, orc, "fused rollout")
    assert np.array_equal(rsum.cpu().numpy(), ref)
    dev.rollout(5, _abi.POLICY_RANDOM); orc.rollout(5, _abi.POLICY_RANDOM)
    same_state(dev, orc, "random-policy rollout")


@pytest.mark.gpu
def test_level_curriculum_without_terrain_and_task_surface():
    """terrain.curriculum from TASK_CFG without any terrain (what the reference's config can express):
    all arenas level 0; through the VecEnv surface."""
    from options_in_simulation_amd.vecenv import VecEnv
    cfg = default_task_cfg()
    cfg["task"]["env"]["terrain"]["curriculum"] = True
    N = 70
    env = VecEnv(cfg, num_envs=N, variant="KAMMA_medium", seed=3)
    orc = OracleEnv(build_config(cfg, variant="KAMMA_medium", num_envs=N, seed=3))
    obs, _ = env.reset()
    orc.reset(3); orc.step(np.zeros((N, 4), np.float32))
    assert np.array_equal(obs.cpu().numpy(), orc.obs)
    g = torch.Generator().manual_seed(0)
    for t in range(120):
        a = torch.rand((N, 4), generator=g)
        obs, rew, term, trunc, info = env.step(a.cuda())
        orc.step(a.numpy())
    assert np.array_equal(obs.cpu().numpy(), orc.obs) and np.array_equal(rew.cpu().numpy()[:, 0], orc.rew)
    assert np.array_equal(env.task.commands.cpu().numpy(), orc.buffer(_abi.BUF_COMMANDS))
    assert np.array_equal(env.task.fraction.cpu().numpy(), orc.buffer(_abi.BUF_CMD_FRACTION))
    assert not env.task.commands[:, 0].any() and (env.task.commands[:, 2] > 0).all()
    env.close()


@pytest.mark.gpu
@pytest.mark.parametrize("N,waves", [(64, None), (40000, None), (70000, None), (40000, "2")], ids=["lane-pair", "one-lane", "two-waves", "one-lane-forced-w2"])
@pytest.mark.parametrize("ground", ["flat", "slopes"])
def test_level_curriculum_rollouts_on_the_plane_fold_shells(N, waves, ground, monkeypatch):
    """the per-level command curriculum WITHOUT a terrain — the only one the reference's TASK_CFG can express —
    in fused rollouts: k > 1 makes launch_steps() set bit 9 (the continuation-launch flag), i.e. the six plane FOLD shells
    (step_kernel<1|2, true, 4, flat|general, true>, step_kernel_w2<true, 4, *, true>) do the previous step's command re-draw
    in their prologue. Against the oracle, bit for bit, every state buffer + the reward sums: EXTERNAL and RANDOM policy; an
    arena falling in the FIRST step of a rollout (the next launch's fold must take the fraction re-snapshot from S.sa[0] of
    every arena), another falling one step before a rollout (the stamp of the other parity), a synchronous timeout inside a
    rollout (6-step episodes); flat and after ois_randomize_params with slopes (the general shells)."""
    from options_in_simulation_amd.vecenv import NativeEnv
    if waves:
        monkeypatch.setenv("OIS_STEP_WAVES", waves)
    cfg = build_config(variant="KAMMA_medium", num_envs=N, seed=13, max_sim_steps=300.0, cmd_sampler=_abi.CMD_LEVEL, max_episode_length=7)
    dev, orc = NativeEnv(cfg, "cuda:0"), OracleEnv(cfg)
    if ground == "slopes":
        sl = [-0.3, -0.2, 0.0, 0.2, 0.3]
        dev.randomize_params(0.5, 1.0, sl); orc.randomize_params(0.5, 1.0, sl)
    info = dev.kernel_info()
    assert ("plane=flat" in info) == (ground == "flat") and "+level_command_kernel" in info
    assert info.startswith("step_kernel<2," if N == 64 else ("step_kernel_w2<" if (N > 65536 or waves == "2") else "step_kernel<1,"))
    dev.reset(13); orc.reset(13)
    same_state(dev, orc, "post_reset")
    rng = np.random.default_rng(4)
    up = np.array([0.0, 1.0, 0.0, 0.0], np.float32)

    def fall(i):
        orc.buffer(_abi.BUF_QUAT)[i] = up
        dev.buffer(_abi.BUF_QUAT)[i] = torch.from_numpy(up).cuda()

    # 1. EXTERNAL actions, an arena falling in the first step of the rollout
    fall(N // 3)
    acts = rng.uniform(0, 1, (4, N, 4)).astype(np.float32)
    rs = torch.zeros(N, device="cuda:0")
    dev.rollout(4, _abi.POLICY_EXTERNAL, torch.from_numpy(acts).cuda(), reward_sum=rs)
    ref = orc.rollout(4, _abi.POLICY_EXTERNAL, acts, want_reward_sum=True)
    same_state(dev, orc, "external rollout with a fall in its first step")
    assert np.array_equal(rs.cpu().numpy(), ref) and orc.buffer(_abi.BUF_CMD_FRACTION).any()
    # 2. a single step in which an arena falls, then a RANDOM-policy rollout that crosses the synchronous timeout
    fall(N - 1)
    a = rng.uniform(0, 1, (N, 4)).astype(np.float32)
    dev.step(torch.from_numpy(a).cuda()); orc.step(a)
    same_state(dev, orc, "single step with a fall")
    rs = torch.zeros(N, device="cuda:0")
    dev.rollout(5, _abi.POLICY_RANDOM, reward_sum=rs)
    ref = orc.rollout(5, _abi.POLICY_RANDOM, None, want_reward_sum=True)
    same_state(dev, orc, "random rollout across the timeout")
    assert np.array_equal(rs.cpu().numpy(), ref)
    cnt = orc.buffer(_abi.BUF_COUNTERS)
    assert (cnt[:, 2] >= 2).all(), "every arena has finished an episode inside the rollouts"
    # 3. rollouts == single steps (the fold changes launches, not results)
    blob = dev.save_state()
    acts = rng.uniform(0, 1, (3, N, 4)).astype(np.float32)
    dev.rollout(3, _abi.POLICY_EXTERNAL, torch.from_numpy(acts).cuda())
    fused = {b: dev.buffer(b).clone() for b in STATE}
    dev.load_state(blob)
    for k in range(3):
        dev.step(torch.from_numpy(acts[k]).cuda())
    for b in STATE:
        assert torch.equal(dev.buffer(b), fused[b]), b
    orc.rollout(3, _abi.POLICY_EXTERNAL, acts)
    same_state(dev, orc, "single steps after load_state")
    dev.close()


def test_eval_sampling_refuses_a_curriculum_terrain():
    """the *_eval files test `evalsampling` before `curriculum` (KAMMA_eval.py:544): with both set the per-step re-draw of
    KAMMA.py:719-725 would hand out the eval ramp, which nothing here implements: refused, not silently different"""
    from options_in_simulation_amd import default_task_cfg
    cfg = default_task_cfg()
    cfg["task"]["env"]["terrain"]["curriculum"] = True
    with pytest.raises(ValueError, match="EvalSampling"):
        build_config(task_cfg=cfg, variant="KAMMA_eval")
    env = OracleEnv(build_config(variant="KAMMA_eval", num_envs=4, seed=1))
    T = tg.Terrain({"curriculum": True, "numLevels": 3, "numTerrains": 2, "mapLength": 8.0, "mapWidth": 8.0}, 4, seed=0)
    with pytest.raises(RuntimeError, match="EvalSampling"):
        env.attach_terrain(T, np.zeros((4, 2), np.int32), curriculum=True)
