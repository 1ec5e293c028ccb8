"""-m gpu: golden trajectories through the HIP kernel, the VecEnv drop-in surface the
reference's trainers use, and size-independent properties at BASELINE.json's full sizes."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

from options_in_simulation_amd import _abi, build_config, default_task_cfg  # noqa: E402

G = os.path.join(os.path.dirname(__file__), "golden")


class _Dev:
    """adapter: NativeEnv with numpy actions (what tests/golden_replay.py drives)"""

    def __init__(self, cfg):
        from options_in_simulation_amd.vecenv import NativeEnv
        self.env = NativeEnv(cfg, "cuda:0")

    def reset(self, seed): self.env.reset(seed)
    def step(self, a): self.env.step(torch.from_numpy(a).cuda())
    def buffer(self, bid): return self.env.buffer(bid)
    def episode_info(self): return self.env.episode_info()
    def stats(self): return self.env.stats()


@pytest.mark.parametrize("name", ["traj_config1_zero_torque", "traj_random_torque", "traj_expert_only", "traj_kamma_switch"])
def test_kernel_reproduces_golden_trajectories(name):
    """committed oracle trajectories (tests/golden/*.npz): 1e-5 relative (north_star), masks and
    counters bit-exact. 1000-step runs: the kernel is bit-identical in practice, the tolerance
    is the contract."""
    from tests.golden_replay import replay
    d = np.load(os.path.join(G, name + ".npz"))
    replay(d, name, _Dev, to_numpy=lambda t: t.detach().cpu().numpy(), exact=False, rtol=1e-5, atol=1e-6)


def test_vecenv_drop_in_surface():
    """what Trainer.single_agent_train touches (src/my_trainers/base_KA_DDPG.py:153-293)"""
    from options_in_simulation_amd import VecEnv
    N = 64
    cfg = default_task_cfg()
    cfg["task"]["env"]["numEnvs"] = N
    env = VecEnv(cfg, device="cuda:0", variant="KAMMA_medium", seed=42)
    assert env.num_envs == N and env.num_agents == 1 and env.device.type == "cuda"
    assert env.observation_space.shape == (6,) and env.action_space.shape == (4,)
    assert float(env.action_space.low.min()) == 0.0 and float(env.observation_space.high.max()) == 1.0
    states, infos = env.reset()
    assert states.shape == (N, 6) and states.dtype == torch.float32 and infos == {}
    s2, _ = env.reset()
    assert s2 is states  # skrl's wrapper only resets once, later calls return the cached obs
    assert (env.task.episode_count == 1).all() and (env.task.sim_steps == 1).all()  # reset() = one zero-action step
    for t in range(3):
        actions = torch.rand((N, 4), device="cuda")
        nxt, rew, term, trunc, infos = env.step(actions)
        assert nxt.shape == (N, 6) and rew.shape == (N, 1) and term.shape == (N, 1) and trunc.shape == (N, 1)
        assert term.dtype == torch.bool and trunc.dtype == torch.bool and not trunc.any()
        assert nxt.data_ptr() != states.data_ptr()  # outputs are copies: `states = next_states` stays valid
        states = nxt
    # env.task.* attributes read by the trainer (base_KA_DDPG.py:184-212)
    oc = env.task.observed_components
    assert len(oc) == 26 and all(k.startswith("env0_") for k in oc)
    assert oc["env0_episode_count"] == 1 and oc["env0_desired_omega"] == 0.0
    assert oc["env0_policy_torque_fl"] == pytest.approx(3.0 * float(actions[0, 0]), rel=1e-6)
    executed = env.task.wheel_torqs / env.task.action_scale
    assert executed.shape == (N, 4) and env.task.action_scale == 3.0
    # infos["episode"]: dict of 0-dim tensors (base_KA_DDPG.py:259-262), keys of KAMMA.py:480-488
    ep = infos["episode"]
    keys = set(ep.keys())
    assert len(keys) == 22
    assert {"rew_Final reward", "rew_r1/Final reward", "rew_Dense/Final reward", "rew_Dense reward/Final reward",
            "gamma_1 action assist", "gamma_2 reward assist", "terrain_level"} <= keys
    assert all(isinstance(v, torch.Tensor) and v.numel() == 1 for v in ep.values())
    assert float(ep["gamma_1 action assist"]) == 1.0 and float(ep["rew_Dense reward/Final reward"]) == 0.0
    env.render()
    env.close()


def test_ddpg_variant_spaces_and_eval_keys():
    from options_in_simulation_amd import VecEnv
    env = VecEnv(num_envs=8, device="cuda:0", variant="KA_DDPG_2D")
    assert env.action_space.shape == (2,) and float(env.action_space.low.min()) == -1.0
    env.reset()
    env.step(torch.zeros((8, 2), device="cuda"))
    env.close()
    ev = VecEnv(num_envs=8, device="cuda:0", variant="KAMMA_eval")
    ev.reset()
    _, _, _, _, infos = ev.step(torch.rand((8, 4), device="cuda"))
    assert {"rew_Smoothness", "rew_Tracking error", "rew_Desired velocity"} <= set(infos["episode"].keys())
    ev.close()


@pytest.mark.parametrize("N", [65536, 100003])
def test_full_size_properties(N):
    """BASELINE.json sizes (65 536 arenas; a ragged prime-ish size): determinism, fused rollout =
    single steps, counters, finite outputs, and a strided sample against the oracle."""
    from options_in_simulation_amd.vecenv import NativeEnv
    from oracle.binding import OracleEnv
    cfg = build_config(variant="KAMMA_medium", num_envs=N, seed=13, max_sim_steps=60.0)
    a, b = NativeEnv(cfg, "cuda:0"), NativeEnv(cfg, "cuda:0")
    a.reset(13); b.reset(13)
    K = 101  # crosses the synchronous timeout at step 99
    a.rollout(K, _abi.POLICY_RANDOM)
    for _ in range(K):
        b.rollout(1, _abi.POLICY_RANDOM)
    torch.cuda.synchronize()
    for bid in (_abi.BUF_POSE, _abi.BUF_QUAT, _abi.BUF_LINVEL, _abi.BUF_ANGVEL, _abi.BUF_COUNTERS, _abi.BUF_SUMS_B,
                _abi.BUF_OBS, _abi.BUF_REW, _abi.BUF_DONE, _abi.BUF_COMMANDS):
        assert torch.equal(a.buffer(bid), b.buffer(bid)), f"fused rollout != {K} single launches in buffer {bid}"
    cnt = a.buffer(_abi.BUF_COUNTERS).cpu().numpy()
    assert (cnt[:, 1] == K).all() and (cnt[:, 2] == 2).all() and (cnt[:, 0] == K - 99).all()
    assert torch.isfinite(a.buffer(_abi.BUF_OBS)).all() and torch.isfinite(a.buffer(_abi.BUF_REW)).all()
    q = a.buffer(_abi.BUF_QUAT)
    assert (q.norm(dim=1) - 1).abs().max() < 1e-5
    st = a.stats()
    assert st.episodes == N and st.length_sum == 99.0 * N and st.steps == float(N) * K
    # strided sample vs the oracle: arena i of the big env == a 1-arena oracle env at arena_offset i
    ids = list(range(0, N, N // 7))[:7] + [N - 1]
    pose = a.buffer(_abi.BUF_POSE).cpu().numpy()
    obs = a.buffer(_abi.BUF_OBS).cpu().numpy()
    for i in ids:
        o = OracleEnv(build_config(variant="KAMMA_medium", num_envs=1, seed=13, max_sim_steps=60.0, arena_offset=i))
        o.reset(13)
        o.rollout(K, _abi.POLICY_RANDOM)
        np.testing.assert_allclose(pose[i], o.buffer(_abi.BUF_POSE)[0], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(obs[i], o.obs[0], rtol=1e-5, atol=1e-6)
    # extras of the synchronous reset: in-order group sums are deterministic across the two envs
    assert torch.equal(a.episode_info(), b.episode_info())
    assert float(a.episode_info()[_abi.EI_COUNT]) == N


def test_config2_in_kernel_random_policy_equals_host_generated_actions():
    """BASELINE.json configs[1]: 4 096 arenas, gamma = 0, actions ~ U[0,1)^4 from Philox(seed, arena,
    step) generated in the kernel (policy = RANDOM) — and the same values generated on the host, handed over as EXTERNAL
    actions: identical runs. The host side is the published Philox4x32-10 (the oracle's, pinned by Random123's known
    answers): counter (arena lo, arena hi, step, purpose 3), key = seed, u = (x >> 8) * 2^-24."""
    import ctypes as C
    from options_in_simulation_amd.vecenv import NativeEnv
    from oracle.binding import lib
    N, K, seed = 4096, 24, 42
    kw = dict(variant="KAMMA_medium", num_envs=N, seed=seed, gamma1_mode=_abi.GAMMA_CONST, gamma1_const=0.0,
              gamma2_mode=_abi.GAMMA_CONST, gamma2_const=0.0)
    a, b = NativeEnv(build_config(**kw), "cuda:0"), NativeEnv(build_config(**kw), "cuda:0")
    a.reset(seed); b.reset(seed)
    L = lib()
    acts = np.zeros((K, N, 4), np.float32)
    ctr, key, out = (C.c_uint32 * 4)(), (C.c_uint32 * 2)(seed & 0xffffffff, seed >> 32), (C.c_uint32 * 4)()
    for s in range(K):
        for i in range(N):
            ctr[0], ctr[1], ctr[2], ctr[3] = i, 0, s, 3
            L.orc_philox4x32_10(ctr, key, out)
            acts[s, i] = [(x >> 8) * 2.0 ** -24 for x in out]
    assert 0.0 <= acts.min() and acts.max() < 1.0 and abs(acts.mean() - 0.5) < 0.01
    ra, rb = torch.zeros(N, device="cuda"), torch.zeros(N, device="cuda")
    a.rollout(K, _abi.POLICY_RANDOM, reward_sum=ra)
    b.rollout(K, _abi.POLICY_EXTERNAL, torch.from_numpy(acts).cuda(), reward_sum=rb)
    torch.cuda.synchronize()
    for bid in (_abi.BUF_POSE, _abi.BUF_QUAT, _abi.BUF_LINVEL, _abi.BUF_ANGVEL, _abi.BUF_WHEEL_VEL, _abi.BUF_WHEEL_TORQS,
                _abi.BUF_SUMS_A, _abi.BUF_SUMS_B, _abi.BUF_OBS, _abi.BUF_REW, _abi.BUF_COUNTERS):
        assert torch.equal(a.buffer(bid), b.buffer(bid)), bid
    assert torch.equal(ra, rb)
    wt = a.buffer(_abi.BUF_WHEEL_TORQS).cpu().numpy()     # gamma = 0: the executed torques ARE the policy's, a * 3 (clip at 4 idle)
    assert np.array_equal(wt, acts[K - 1] * np.float32(3.0))


def test_config5_partition_at_full_size_on_one_gpu():
    """BASELINE.json configs[4]: 524 288 arenas as 8 shards of 65 536 (arena_offset = rank * 65 536). One GPU here, so:
    the whole range as ONE environment (two waves per SIMD: the 256-register shell) against shards 0, 3 and 7 as
    environments of their own (one lane per arena, default registers) — every buffer of a shard bit-identical to its
    slice of the whole, across the synchronous timeout; and the shards' statistics merge (dist.merge_stats) to the whole's."""
    from options_in_simulation_amd.dist import merge_stats
    from options_in_simulation_amd.vecenv import NativeEnv
    per, world = 65536, 8
    kw = dict(variant="KAMMA_medium", seed=19, max_sim_steps=200.0)
    whole = NativeEnv(build_config(num_envs=per * world, **kw), "cuda:0")
    whole.reset(19)
    K = 100
    whole.rollout(K, _abi.POLICY_RANDOM)
    parts = []
    for r in range(world):
        sh = NativeEnv(build_config(num_envs=per, arena_offset=r * per, **kw), "cuda:0")
        sh.reset(19)
        sh.rollout(K, _abi.POLICY_RANDOM)
        torch.cuda.synchronize()
        if r in (0, 3, 7):
            for bid in (_abi.BUF_POSE, _abi.BUF_QUAT, _abi.BUF_LINVEL, _abi.BUF_ANGVEL, _abi.BUF_WHEEL_VEL, _abi.BUF_COMMANDS,
                        _abi.BUF_COUNTERS, _abi.BUF_SUMS_A, _abi.BUF_SUMS_B, _abi.BUF_OBS, _abi.BUF_REW, _abi.BUF_DONE):
                assert torch.equal(sh.buffer(bid), whole.buffer(bid)[r * per:(r + 1) * per]), (r, bid)
        parts.append(sh.stats())
        sh.close()
    m, w = merge_stats(parts), whole.stats()
    assert m["episodes"] == w.episodes == per * world and m["steps"] == w.steps
    assert m["return_min"] == w.return_min and m["return_max"] == w.return_max
    assert m["return_mean"] == pytest.approx(w.return_sum / w.episodes, rel=1e-12)
    assert m["return_std"] == pytest.approx((w.return_sqsum / w.episodes - (w.return_sum / w.episodes) ** 2) ** 0.5, rel=1e-9)


def test_capture_with_a_transition_memory_attached_is_refused():
    """graph replays do not advance the host-side replay cursor — so such a capture is refused
    instead of silently writing one memory row over and over"""
    from options_in_simulation_amd.vecenv import NativeEnv
    N = 256
    env = NativeEnv(build_config(variant="KAMMA_medium", num_envs=N, seed=1), "cuda:0")
    env.reset(1)
    env.attach_replay(8)
    a = torch.rand((N, 4), device="cuda")
    env.step(a)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        with pytest.raises(RuntimeError, match="transition memory"):
            with torch.cuda.graph(g, stream=s):
                env.step(a)
    torch.cuda.synchronize()
    env.detach_replay()
    env.step(a)  # and the environment is still usable
    torch.cuda.synchronize()
    assert env.replay_cursor() == (0, False)


def test_push_robots_matches_the_oracle_and_only_touches_planar_velocity():
    """push_robots (KAMMA.py:729-733): base_velocities[:, 0:2] ~ U(-1, 1) for every arena, nothing else"""
    from options_in_simulation_amd.vecenv import VecEnv
    from oracle.binding import OracleEnv
    N = 300
    env = VecEnv(num_envs=N, device="cuda:0", variant="KAMMA_medium", seed=9)
    orc = OracleEnv(build_config(variant="KAMMA_medium", num_envs=N, seed=9))
    env.reset(); orc.reset(9); orc.step(np.zeros((N, 4), np.float32))
    a = np.full((N, 4), 0.3, np.float32)
    for _ in range(3):
        env.step(torch.from_numpy(a).cuda()); orc.step(a)
    before = {b: env.task.native.buffer(b).clone() for b in (_abi.BUF_POSE, _abi.BUF_QUAT, _abi.BUF_ANGVEL, _abi.BUF_WHEEL_VEL)}
    vz = env.task.native.buffer(_abi.BUF_LINVEL)[:, 2:].clone()
    env.task.push_robots(); orc.push_robots()
    env.task.refresh_body_state_tensors(); env.task.refresh_dof_state_tensors()
    torch.cuda.synchronize()
    lv = env.task.native.buffer(_abi.BUF_LINVEL)
    assert np.array_equal(lv.cpu().numpy(), orc.buffer(_abi.BUF_LINVEL))
    assert float(lv[:, :2].abs().max()) <= 1.0 and float(lv[:, :2].std()) > 0.4 and torch.equal(lv[:, 2:], vz)
    for b, v in before.items():
        assert torch.equal(env.task.native.buffer(b), v)
    assert torch.equal(env.task.base_velocities[:, :2], lv[:, :2])
    env.step(torch.from_numpy(a).cuda()); orc.step(a)   # and the pushed state steps identically
    assert np.array_equal(env.task.native.buffer(_abi.BUF_POSE).cpu().numpy(), orc.buffer(_abi.BUF_POSE))
    env.close()


def _blocks(N, width=64):
    """contiguous arena blocks spread over [0, N): the oracle steps them as small envs at arena_offset"""
    starts = sorted({0, (N // 3) // 16 * 16 + 5, (2 * N // 3) // 16 * 16 + 11, N - width})
    return [(s0, min(width, N - s0)) for s0 in starts]


@pytest.mark.parametrize("domain_rand", [False, True], ids=["config3_actions", "config4_domain_rand"])
def test_external_actions_at_full_size_against_the_oracle(domain_rand):
    """BASELINE.json configs[2] / configs[3] through their actual entry point at the stated size:
    65 536 arenas, EXTERNAL actions (ois_step, what the actor loop calls), gamma decaying so that the
    policy's torques are executed on part of the steps; with domain_rand also ois_randomize_params at
    this size. Four 64-arena blocks across the range are stepped by the oracle (arena_offset = block
    start, same per-step actions): every state buffer bit-identical, across the synchronous timeout."""
    from options_in_simulation_amd.vecenv import NativeEnv
    from oracle.binding import OracleEnv
    N = 65536
    kw = dict(variant="KAMMA_medium", seed=17, max_sim_steps=150.0)
    dev = NativeEnv(build_config(num_envs=N, **kw), "cuda:0")
    blocks = _blocks(N)
    orcs = [OracleEnv(build_config(num_envs=n, arena_offset=s0, **kw)) for s0, n in blocks]
    slopes = [-0.3, -0.2, 0.0, 0.2, 0.3]  # origin_terrain_generator.py:92
    if domain_rand:
        dev.randomize_params(0.5, 1.0, slopes)
        for o in orcs:
            o.randomize_params(0.5, 1.0, slopes)
    dev.reset(17)
    for o in orcs:
        o.reset(17)
    bufs = [_abi.BUF_POSE, _abi.BUF_QUAT, _abi.BUF_LINVEL, _abi.BUF_ANGVEL, _abi.BUF_WHEEL_VEL, _abi.BUF_COMMANDS,
            _abi.BUF_COUNTERS, _abi.BUF_SUMS_A, _abi.BUF_SUMS_B, _abi.BUF_LAMBDA_N, _abi.BUF_LAMBDA_L, _abi.BUF_LAMBDA_T,
            _abi.BUF_OBS, _abi.BUF_REW, _abi.BUF_DONE, _abi.BUF_TIMEOUT, _abi.BUF_WHEEL_TORQS, _abi.BUF_PARAMS]

    def check(where):
        torch.cuda.synchronize()
        for (s0, n), o in zip(blocks, orcs):
            for bid in bufs:
                got = dev.buffer(bid)[s0:s0 + n].cpu().numpy()
                got = got.astype(np.uint8) if got.dtype == np.bool_ else got
                assert np.array_equal(got, o.buffer(bid)), f"{where}: arenas {s0}..{s0 + n}: buffer {bid} not bit-identical"
    check("post_reset")
    g = torch.Generator(device="cuda").manual_seed(5)
    policy_steps = 0
    for t in range(104):
        a = torch.rand((N, 4), device="cuda", generator=g)
        dev.step(a)
        for (s0, n), o in zip(blocks, orcs):
            o.step(a[s0:s0 + n].cpu().numpy())
        if t in (0, 60, 98, 99, 103):
            check(f"step {t}")
    wt = dev.buffer(_abi.BUF_WHEEL_TORQS)
    assert torch.isfinite(dev.buffer(_abi.BUF_OBS)).all() and wt.abs().max() <= 4.0
    st = dev.stats()
    assert st.episodes >= N and st.steps == float(N) * 104
    if domain_rand:
        p = dev.buffer(_abi.BUF_PARAMS)
        assert 0.5 <= float(p[:, 0].min()) and float(p[:, 0].max()) <= 1.0 and len(torch.unique(p[:, 3])) == 3  # |slope| in {0, 0.2, 0.3}


def test_step_is_capturable_in_a_hip_graph():
    """policy + ois_step recorded once with torch.cuda.graph and replayed: the library only
    enqueues on the caller's stream (no allocation, no synchronisation in ois_step), so the
    launch is capturable; the replays reproduce the eager run bit for bit."""
    from options_in_simulation_amd.vecenv import NativeEnv
    N = 1000
    cfg = build_config(variant="KA_DDPG_4D", num_envs=N, seed=6)
    eager, graphed = NativeEnv(cfg, "cuda:0"), NativeEnv(cfg, "cuda:0")
    eager.reset(6); graphed.reset(6)

    def policy(obs):
        return torch.tanh(obs[:, :4] * 0.5 + obs[:, 2:6])

    for _ in range(7):
        eager.step(policy(eager.buffer(_abi.BUF_OBS)))
    obs = graphed.buffer(_abi.BUF_OBS)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        graphed.step(policy(obs))          # warm-up step outside the graph (step 1)
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        graphed.step(policy(obs))          # captured, not executed
    for _ in range(6):
        g.replay()                         # steps 2..7
    torch.cuda.synchronize()
    for bid in (_abi.BUF_OBS, _abi.BUF_REW, _abi.BUF_POSE, _abi.BUF_QUAT, _abi.BUF_COUNTERS, _abi.BUF_SUMS_B):
        assert torch.equal(eager.buffer(bid), graphed.buffer(bid)), bid
    # the step's host logs around a graph: with ois_logs_enable on, a CAPTURED step does not enqueue the publishing kernel (its
    # sequence number advances per call, not per replay) — the capture still works, and after replays the fetch finalises
    # the current state itself once the host caches are told that the state moved (NativeEnv.state_changed)
    logged = NativeEnv(cfg, "cuda:0")
    logged.reset(6)
    logged.step(policy(logged.buffer(_abi.BUF_OBS)))       # step 1
    first = logged.host_logs()[1].copy()                   # switches ois_logs_enable on
    assert logged._logs_eager
    obs2 = logged.buffer(_abi.BUF_OBS)
    g2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g2):
        logged.step(policy(obs2))
    for _ in range(6):
        g2.replay()
    torch.cuda.synchronize()
    logged.state_changed()
    ei, diag = logged.host_logs()[:2]
    assert np.array_equal(diag, eager.env0_diag(direct=True)) and not np.array_equal(diag, first)
    assert np.array_equal(ei, eager.episode_info().cpu().numpy(), equal_nan=True)


def test_ppo_example_runs():
    import importlib.util
    spec = importlib.util.spec_from_file_location("ppo_rollout", os.path.join(os.path.dirname(os.path.dirname(__file__)),
                                                                            "examples", "ppo_rollout.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.main(["--num-envs", "256", "--rollouts", "8", "--updates", "2"]) == 0


def test_example_loop_runs(tmp_path):
    """examples/kaddpg_loop.py: env + device memory + sampling + tracker end to end"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("kaddpg_loop", os.path.join(os.path.dirname(os.path.dirname(__file__)),
                                                                            "examples", "kaddpg_loop.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.main(["--num-envs", "512", "--timesteps", "120", "--memory-size", "8", "--batch-size", "1024",
                     "--logdir", str(tmp_path)]) == 0
    from options_in_simulation_amd.tracking import timeseries_from_events
    ev = [f for f in os.listdir(tmp_path) if f.startswith("events.out.tfevents")]
    steps, cols = timeseries_from_events(os.path.join(tmp_path, ev[0]))
    assert steps == [30, 60, 90, 120] and np.isfinite(cols["current_v"]).all()
    assert os.path.exists(os.path.join(tmp_path, "reward_components_env0_timeseries.csv"))


def test_combined_host_logs_equal_the_separate_reads():
    """task.observed_components and infos['episode'] come out of ONE published host block (NativeEnv.host_logs ->
    ois_logs_fetch; from the first read on the step publishes it itself): same values as the two separate entry
    points ois_episode_info + ois_env0_diag on the device buffers"""
    from options_in_simulation_amd import VecEnv
    env = VecEnv(num_envs=300, device="cuda:0", variant="KAMMA_eval", seed=9)
    env.reset()
    for t in range(120):
        _, _, _, _, infos = env.step(torch.rand((300, 4), device="cuda"))
        if t < 100 and t % 7:
            continue
        oc = dict(env.task.observed_components)          # first: diag path (combined after the first infos read)
        ep = {k: float(v) for k, v in infos["episode"].items()}
        n = env.task.native
        direct_info = n.episode_info().cpu().numpy()
        direct_diag = n.env0_diag(direct=True)           # the plain ois_env0_diag copy of the device buffer
        assert np.array_equal(n.host_logs()[1], direct_diag, equal_nan=True)
        assert np.array_equal(n.host_logs()[0], direct_info, equal_nan=True)
        assert oc["env0_desired_v"] == float(direct_diag[0]) and oc["env0_episode_count"] == int(direct_diag[8])
        assert oc["env0_torque_apl_rr"] == float(direct_diag[12])
        for slot, key in _abi.EPISODE_INFO_KEYS.items():
            a, b = ep[key], float(direct_info[slot])
            assert a == b or (np.isnan(a) and np.isnan(b)), key
        assert all(v.device.type == "cpu" for v in infos["episode"].values())
        assert list(infos["episode"]) == [k for k, _ in _abi.episode_info_layout(True, False)]   # the reference's dict order (KAMMA_eval)
    n = env.task.native
    assert n._logs_eager                                 # the loop read its logs: ois_step publishes them itself
    for _ in range(18):                                  # ... and stops paying for that once nobody reads them
        env.step(torch.rand((300, 4), device="cuda"))
    assert not n._logs_eager
    _, _, _, _, infos = env.step(torch.rand((300, 4), device="cuda"))
    ep = {k: float(v) for k, v in infos["episode"].items()}
    direct_info = n.episode_info().cpu().numpy()
    for slot, key in _abi.EPISODE_INFO_KEYS.items():
        assert ep[key] == float(direct_info[slot]) or (np.isnan(ep[key]) and np.isnan(direct_info[slot])), key
    assert n._logs_eager
    env.close()


def test_c_abi_from_plain_c(tmp_path):
    """tests/c_abi/c_abi_smoke.c: the boundary of include/ois.h driven from a C11 program built
    with gcc (no Python, no torch types): hipMalloc'ed workspace and actions, step / rollout /
    buffers / stats / state blob, error codes."""
    import shutil
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    libdir = os.path.join(root, "options_in_simulation_amd", "lib")
    if not (shutil.which("gcc") and os.path.exists("/opt/rocm/include/hip/hip_runtime_api.h")):
        pytest.skip("gcc or the HIP headers are missing")
    exe = str(tmp_path / "c_abi_smoke")
    subprocess.run(["gcc", "-std=c11", "-O1", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-I" + os.path.join(root, "include"),
                    "-o", exe, os.path.join(root, "tests", "c_abi", "c_abi_smoke.c"), "-L" + libdir, "-lois_hip", "-L/opt/rocm/lib",
                    "-lamdhip64", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-lm"], check=True, capture_output=True)
    means = []
    for device in range(torch.cuda.device_count()):  # every ordinal this box has: one today, eight on the scaling node
        res = subprocess.run([exe, str(device)], capture_output=True, text=True, timeout=120)
        assert res.returncode == 0, res.stdout + res.stderr
        assert "c abi ok: 1000 arenas, 1000 episodes" in res.stdout and res.stdout.strip().endswith(f"on device {device}")
        means.append(res.stdout.split("mean return")[1].split()[0])
    assert len(set(means)) == 1  # the same numbers on every GPU


# ---------------------------------------------------------------------------------------------------------------------
# the caller's side of the boundary as the REFERENCE's trainer exercised it (tests/golden/refexec_trainer_trace.json, made by
# tests/golden/make_golden_trainer.py: Trainer.single_agent_train / single_agent_eval of src/my_trainers/base_KA_DDPG.py
# executed against a recording environment built from the reference's own task statements)
def _summary(v):
    if isinstance(v, torch.Tensor):
        return {"type": "tensor", "shape": list(v.shape), "dtype": str(v.dtype).replace("torch.", "")}
    if isinstance(v, bool):
        return {"type": "bool"}
    if isinstance(v, int):
        return {"type": "int"}
    if isinstance(v, float):
        return {"type": "float"}
    if isinstance(v, dict):
        return {"type": "dict", "keys": sorted(v), "values": {k: _summary(v[k]) for k in sorted(v)}}
    if isinstance(v, (tuple, list)):
        return {"type": "tuple", "items": [_summary(x) for x in v]}
    if v is None:
        return {"type": "none"}
    return {"type": type(v).__name__}


def _trace_runs():
    import json
    return json.load(open(os.path.join(G, "refexec_trainer_trace.json")))["runs"]


@pytest.mark.parametrize("index", range(4))
def test_trainer_trace_replays_on_the_device(index):
    """Every attribute the reference's trainer read resolves on the product's VecEnv, every key exists, every call takes the
    recorded argument and returns the recorded structure (types, shapes, dtypes, key lists), in the recorded order — including
    the 26 `observed_components` reads (:184-209), `wheel_torqs / action_scale` (:212), the `infos["episode"]` loop (:259-262:
    one `Info / <key>` tag per key) and, with one arena, the second `env.reset()` after the step that ended the episode
    (:268-275, twice: the reference has the block twice), which must hand back the cached observation as skrl's wrapper does."""
    from options_in_simulation_amd import VecEnv
    run = _trace_runs()[index]
    N = run["num_envs"]
    env = VecEnv(num_envs=N, device="cuda:0", variant=run["variant"], seed=3)
    events = [e for b in run["blocks"] for _ in range(b["repeat"]) for e in b["events"]] + run["tail"]
    g = torch.Generator(device="cuda").manual_seed(0)
    cur = {}          # what the trainer's local variables hold
    tags, resets, steps = [], 0, 0

    def check_tags():   # one `Info / <key>` tag per key of the step's infos["episode"] (logged by both copies of the block)
        if "step_infos" in cur:
            assert set(tags) == {"Info / " + k for k in cur["step_infos"].get("episode", {})}, (steps, sorted(tags))

    def resolve(path):
        obj = env
        for name in path.split(".")[1:]:
            obj = getattr(obj, name)
        return obj
    for n_ev, e in enumerate(events):
        where = f"event {n_ev}: {e.get('op')} {e.get('path', e.get('name'))} {e.get('key', '')}"
        if e["op"] == "getattr":
            v = resolve(e["path"])
            want = e["result"]
            if want["type"] in ("tensor", "int", "float", "bool"):
                got = _summary(v)
                assert got == {k: want[k] for k in got}, (where, got, want)
                if "value" in want:
                    assert v == want["value"], where
            elif want["type"] == "dict":
                assert isinstance(v, dict), where
            cur[e["path"]] = v
        elif e["op"] == "getitem":
            v = resolve(e["path"])[e["key"]]
            assert _summary(v) == e["result"], (where, _summary(v), e["result"])
        elif e["op"] == "call" and e["path"] == "env.reset":
            states, infos = env.reset()
            assert _summary((states, infos)) == e["result"], (where, _summary((states, infos)), e["result"])
            if resets:   # not the first call: skrl's wrapper returns its cached observation (the last step's)
                assert bool(cur["terminated"].any()) and torch.equal(states, cur["next_states"]), where
            cur["states"], cur["infos"] = states, infos
            resets += 1
        elif e["op"] == "call" and e["path"] == "env.step":
            (a,) = e["args"]
            lo = float(env.action_space.low.min())
            actions = (1.0 - lo) * torch.rand(a["shape"], device="cuda", generator=g) + lo
            assert _summary(actions) == a
            check_tags()
            out = env.step(actions)
            got = _summary(out)
            assert got == e["result"], (where, got, e["result"])
            cur["next_states"], cur["rewards"], cur["terminated"], cur["truncated"], cur["infos"] = out
            cur["step_infos"] = out[4]
            tags.clear()
            steps += 1
        elif e["op"] == "call" and e["path"] == "env.render":
            env.render()
        elif e["op"] == "agent" and e["name"] == "act":
            assert _summary(cur["states"]) == e["states"], where
        elif e["op"] == "agent" and e["name"] == "record_transition":
            kw = e["kwargs"]
            if run["method"] == "single_agent_train":   # :212 the executed torques replace the agent's action
                executed = cur["env.task.wheel_torqs"] / cur["env.task.action_scale"]
                assert _summary(executed) == kw["actions"], where
            for k in ("states", "rewards", "next_states", "terminated", "truncated", "infos"):
                assert _summary(cur[k]) == kw[k], (where, k, _summary(cur[k]), kw[k])
            cur["states_before"] = cur["states"]
        elif e["op"] == "agent" and e["name"] == "track_data" and e["tag"].startswith("Info / "):
            tags.append(e["tag"])
            v = cur["infos"]["episode"][e["tag"][len("Info / "):]]
            assert isinstance(v, torch.Tensor) and v.numel() == 1 and isinstance(v.item(), float), where   # :261-262
        elif e["op"] == "agent" and e["name"] in ("post_interaction", "base.post_interaction"):
            if N > 1 or not bool(cur["terminated"].any()):
                cur["states"] = cur["next_states"]      # :266-267 / :276
    check_tags()
    assert resets == run["env_reset_calls"] and steps == run["timesteps"]
    env.close()


def test_flat_plane_shell_follows_the_params_and_the_state_blob():
    """ois_kernel_info: the flat-plane shells run exactly while every arena stands on n = (0, 0, 1) — slopes from
    ois_randomize_params leave them, friction alone does not, a state blob carries the flag — and a flat run equals the same
    run forced through the general kernels (OIS_FLAT=0 is read at ois_create) value for value."""
    from options_in_simulation_amd.vecenv import NativeEnv
    N = 40000
    cfg = build_config(variant="KAMMA_medium", num_envs=N, seed=3)
    a = NativeEnv(cfg, "cuda:0")
    assert a.kernel_info().startswith("step_kernel<1,1,4,1>") and "plane=flat" in a.kernel_info()
    os.environ["OIS_FLAT"] = "0"
    try:
        b = NativeEnv(cfg, "cuda:0")
    finally:
        del os.environ["OIS_FLAT"]
    assert b.kernel_info().startswith("step_kernel<1,1,4,0>") and "plane=per-arena" in b.kernel_info()
    a.reset(3); b.reset(3)
    a.rollout(130, _abi.POLICY_RANDOM); b.rollout(130, _abi.POLICY_RANDOM)
    for bid in (_abi.BUF_POSE, _abi.BUF_QUAT, _abi.BUF_LINVEL, _abi.BUF_ANGVEL, _abi.BUF_WHEEL_VEL, _abi.BUF_SUMS_A, _abi.BUF_SUMS_B,
                _abi.BUF_OBS, _abi.BUF_REW, _abi.BUF_COUNTERS, _abi.BUF_COMMANDS):
        assert torch.equal(a.buffer(bid), b.buffer(bid)), bid      # IEEE value equality: -0 == +0 (docs/SPEC.md 3.2)
    blob = a.save_state()
    a.randomize_params(0.5, 1.0, [0.0, 0.2])
    assert "plane=per-arena" in a.kernel_info()
    sloped = a.save_state()
    a.randomize_params(0.5, 1.0)
    assert "plane=flat" in a.kernel_info()                          # friction only
    a.load_state(sloped)
    assert "plane=per-arena" in a.kernel_info()                     # the blob's params rows hold slopes
    a.load_state(blob)
    assert "plane=flat" in a.kernel_info()
    small = NativeEnv(build_config(variant="KAMMA_medium", num_envs=64, seed=3), "cuda:0")
    assert small.kernel_info().startswith("step_kernel<2,1,4,1>")   # a lane pair per arena up to 32 768 arenas
    big = NativeEnv(build_config(variant="KAMMA_medium", num_envs=140000, seed=3), "cuda:0")
    assert big.kernel_info().startswith("step_kernel_w2<1,4,1>")    # two waves per SIMD above 65 536
    for e in (a, b, small, big):
        e.close()


@pytest.mark.gpu
def test_randomize_params_refuses_what_the_contact_model_cannot_take():
    """a negative / non-finite friction turns the friction cone's clamp inside out, a plane that does not face up cannot carry a
    robot: OIS_EINVAL from the library and from the oracle alike, the environment unchanged"""
    from options_in_simulation_amd.vecenv import NativeEnv
    from oracle.binding import OracleEnv
    cfg = build_config(variant="KAMMA_medium", num_envs=8, seed=1)
    dev, orc = NativeEnv(cfg, "cuda:0"), OracleEnv(cfg)
    dev.reset(1); orc.reset(1)
    before = dev.buffer(_abi.BUF_PARAMS).clone()
    for args in ((-0.1, 1.0, []), (1.0, 0.5, []), (0.5, float("nan"), []), (0.5, float("inf"), []), (0.5, 1.0, [0.1, 20.0]), (0.5, 1.0, [float("nan")])):
        for env in (dev, orc):
            with pytest.raises(RuntimeError):
                env.randomize_params(*args)
    torch.cuda.synchronize()
    assert torch.equal(dev.buffer(_abi.BUF_PARAMS), before)
    dev.randomize_params(0.0, 0.0, [10.0]); orc.randomize_params(0.0, 0.0, [10.0])   # the edges of the accepted ranges
    assert np.array_equal(dev.buffer(_abi.BUF_PARAMS).cpu().numpy(), orc.buffer(_abi.BUF_PARAMS))
    dev.close(); orc.close()
