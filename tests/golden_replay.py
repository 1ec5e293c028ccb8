"""Trajectory specs shared by tests/golden/make_golden.py (which records them with the CPU
oracle) and the tests that replay them (oracle regression on CPU, HIP kernel on the GPU)."""
import numpy as np

from options_in_simulation_amd import _abi, build_config

TRAJ_SPECS = {
    # BASELINE.json configs[0]: 1 env, flat, zero torque, 1000 steps; gamma = 0 so that zero
    # torque is what is executed (KAMMA_eval switches)
    "traj_config1_zero_torque": dict(variant="KAMMA_eval", n=1, steps=1000, policy="zero", seed=42, every=1, kw={}),
    # configs[1] in miniature: random torque policy
    "traj_random_torque": dict(variant="KAMMA_eval", n=8, steps=1000, policy="random", seed=42, every=4, kw={}),
    # expert only (gamma = 1 forever): the classical controller of the reference's figure
    "traj_expert_only": dict(variant="KAMMA_g11_g21", n=8, steps=300, policy="zero", seed=7, every=1, kw={}),
    # the shipped training variant with a fast gamma decay so both branches of the switch run
    "traj_kamma_switch": dict(variant="KAMMA_medium", n=16, steps=300, policy="random", seed=3, every=1,
                              kw=dict(max_sim_steps=200.0)),
}

STATE_IDS = [_abi.BUF_POSE, _abi.BUF_QUAT, _abi.BUF_LINVEL, _abi.BUF_ANGVEL, _abi.BUF_WHEEL_VEL, _abi.BUF_COMMANDS,
             _abi.BUF_COUNTERS, _abi.BUF_SUMS_A, _abi.BUF_SUMS_B, _abi.BUF_LAMBDA_N, _abi.BUF_LAMBDA_L,
             _abi.BUF_LAMBDA_T, _abi.BUF_OBS, _abi.BUF_REW, _abi.BUF_DONE, _abi.BUF_TIMEOUT, _abi.BUF_WHEEL_TORQS]


def run_spec(spec, make_env, to_numpy):
    """make_env(cfg) -> object with reset(seed), step(np.ndarray), buffer(id), episode_info(), stats()."""
    cfg = build_config(variant=spec["variant"], num_envs=spec["n"], seed=spec["seed"], **spec["kw"])
    env = make_env(cfg)
    env.reset(spec["seed"])
    rng = np.random.default_rng(spec["seed"])
    n, A = spec["n"], cfg.action_dim
    obs0, rew, dones = [], [], []
    for t in range(spec["steps"]):
        a = np.zeros((n, A), np.float32) if spec["policy"] == "zero" else rng.uniform(0.0, 1.0, (n, A)).astype(np.float32)
        env.step(a)
        if t % spec["every"] == 0:
            obs0.append(to_numpy(env.buffer(_abi.BUF_OBS))[0].copy())
            rew.append(to_numpy(env.buffer(_abi.BUF_REW)).copy())
            dones.append(int(to_numpy(env.buffer(_abi.BUF_DONE)).astype(np.uint8).sum()))
    out = {"obs_env0": np.array(obs0), "rew": np.array(rew), "done_count": np.array(dones, np.int32),
           "episode_info": to_numpy(env.episode_info()).copy()}
    for bid in STATE_IDS:
        arr = to_numpy(env.buffer(bid))
        out[f"buf{bid}"] = (arr.astype(np.uint8) if arr.dtype == np.bool_ else arr).copy()
    s = env.stats()
    out["stats"] = np.array([s.episodes, s.return_sum, s.return_sqsum, s.return_min, s.return_max, s.length_sum, s.steps])
    return out


def replay(fixture, name, make_env, to_numpy, exact, rtol=1e-5, atol=1e-6):
    got = run_spec(TRAJ_SPECS[name], make_env, to_numpy)
    for key, ref in fixture.items():
        g = got[key]
        if ref.dtype.kind in "iub":
            assert np.array_equal(g, ref), f"{name}:{key} must be bit-exact"
        elif exact:
            assert np.array_equal(g, ref, equal_nan=True), f"{name}:{key} is not bit-identical to the fixture"
        else:
            np.testing.assert_allclose(g, ref, rtol=rtol, atol=atol, equal_nan=True, err_msg=f"{name}:{key}")
