"""Pins against tests/golden/refexec_*.npz: vectors made by EXECUTING the reference's own function definitions
(taken out of its files by ast and compiled unchanged, tests/golden/make_golden_ref.py) — calculate_metrics,
get_observations, sample_velocity_command (curriculum and EvalSampling branches), PPO's compute_gae, the terrain
generator families and the Terrain class. Stronger than the transcription fixtures (refmath_*.npz) for the rows they
cover: no line of those functions was restated by hand. Functions whose bodies need omni / skrl helpers stay on
refmath_*.npz."""
import ctypes as C
import hashlib
import os

import numpy as np
import pytest

from options_in_simulation_amd import _abi
from options_in_simulation_amd import terrain as tg
from oracle.binding import fptr, u8ptr

G = os.path.join(os.path.dirname(__file__), "golden")


def f32(x):
    return np.ascontiguousarray(x, dtype=np.float32)


def case(npz, name):
    d = np.load(os.path.join(G, npz))
    return {k.split("__", 1)[1]: d[k] for k in d.files if k.startswith(name + "__")}


@pytest.mark.parametrize("name", ["kamma_t0", "kamma_t15", "ddpg4d", "eval"])
def test_calculate_metrics_against_the_executed_reference(oracle_lib, name):
    """KAMMA.py:767-822 (and the KA_DDPG_4D / KAMMA_eval versions) as the reference computes it, `~timeout_buf` on the
    int64 tensor included. Element-wise chains exact; r3 = torch.sum over the wheels at 2e-6 (summation order)."""
    d = case("refexec_metrics.npz", name)
    n = len(d["rew"])
    prev_lin, prev_ang = f32(d["prev_lin"]).copy(), f32(d["prev_ang"]).copy()
    sums = f32(d["sums_in"][:, :8]).copy()
    lin, ang, rew = (np.zeros(n, np.float32) for _ in range(3))
    oracle_lib.orc_metrics(n, fptr(f32(d["v_forward_projected"])), fptr(f32(d["omega_z"])), fptr(prev_lin), fptr(prev_ang),
                           fptr(f32(d["v_delta"])), fptr(f32(d["omega_delta"])), fptr(f32(d["wheel_torqs"])), fptr(f32(d["guiding"])),
                           fptr(f32(d["gamma2"])), u8ptr(np.ascontiguousarray(d["reset"])), u8ptr(np.ascontiguousarray(d["timeout"])),
                           np.float32(0.1), -25.0, -0.02, -0.09, 0.2, 0.01, float(d["terminal"]),
                           fptr(lin), fptr(ang), fptr(rew), fptr(sums))
    assert np.array_equal(lin, d["lin_acc"]) and np.array_equal(ang, d["ang_acc"])          # :770-771
    assert np.array_equal(prev_lin, d["prev_lin_out"]) and np.array_equal(prev_ang, d["prev_ang_out"])  # :774-775
    np.testing.assert_allclose(rew, d["rew"], rtol=2e-6, atol=2e-6)                           # :798-801
    np.testing.assert_allclose(sums, d["sums_out"][:, :8], rtol=2e-6, atol=2e-5)              # :803-810
    assert ((np.abs(d["v_delta"]) < 0.01) & (np.abs(d["omega_delta"]) < 0.01)).sum() >= 1      # the sparse bonus occurs
    if float(d["terminal"]):
        both = (d["reset"] == 1) & (d["timeout"] == 1)
        assert both.any() and (d["reset"] == 1).sum() > both.sum()
    if d["sums_in"].shape[1] == 9:  # KAMMA_eval.py:840
        sc = np.zeros((n, 4), np.float32); sc[:, 1] = d["sums_in"][:, 8]
        wt = f32(d["wheel_torqs"])
        oracle_lib.orc_eval_sums(n, fptr(wt), fptr(wt), fptr(np.zeros(n, np.float32)), fptr(f32(d["v_delta"])), fptr(sc))
        assert np.array_equal(sc[:, 1], d["sums_out"][:, 8]) and not sc[:, 0].any() and not sc[:, 2].any()


@pytest.mark.parametrize("name", ["kamma", "g1d", "g1d_normal"])
def test_observations_and_env0_diagnostics_against_the_executed_reference(name):
    """KAMMA.py:825-869 / KAMMA_g1d_g21.py:825-872: obs_buf column order, the 26 observed_components (the percentages are
    Python-double arithmetic on .item() values), the env-0 reward guard. The host mirror builds the dict from the 32-float
    block the kernel leaves; here the block is filled as the kernel fills it (csrc/ois_kernels.hip, oracle step_arena)."""
    from options_in_simulation_amd.vecenv import observed_components_from_diag
    d = case("refexec_observations.npz", name)
    obs = np.stack([d["desired_v"], d["desired_omega"], d["v_delta"], d["omega_delta"], d["linear_acc"], d["angular_acc"]], 1)
    assert np.array_equal(obs, d["obs_buf"])  # :833 (the +-1000 clamp is the wrapper's, rl_task.py:99)
    rew0 = np.float32(d["rew_buf"][0])
    if name.startswith("g1d") and abs(float(rew0)) < 1e-12:
        rew0 = np.float32(1e-12)              # KAMMA_g1d_g21.py:838-842, before the dict is built
    assert d["rew_buf_out"][0] == rew0 and np.array_equal(d["rew_buf_out"][1:], d["rew_buf"][1:])
    diag = np.zeros(_abi.ENV0_DIAG_LEN, np.float32)
    diag[0:8] = [d["desired_v"][0], d["current_v"][0], d["desired_omega"][0], d["current_omega"][0], d["v_delta"][0],
                 d["omega_delta"][0], d["linear_acc"][0], d["angular_acc"][0]]
    diag[8] = d["episode_count"][0]
    diag[9:13] = d["torques"][0]
    diag[13], diag[14] = d["ac_left"][0], d["ac_right"][0]
    diag[15:19] = np.float32(3.0) * d["actions"][0]          # pol[k] = a * action_scale in float32
    diag[19] = d["gamma_assist2"][0]
    diag[20:27] = [d["comp_1"][0], d["comp_2"][0], d["comp_3"][0], d["rdense"][0], d["rsparse"][0], d["robs"][0], d["rguide"][0]]
    diag[27] = rew0
    got = observed_components_from_diag(diag)
    keys = [str(k) for k in d["component_keys"]]
    assert list(got) == keys                                  # same 26 keys in the same order
    for k, ref in zip(keys, d["component_values"]):
        if "policy_torque" in k:  # action_scale * actions.item(): a double product there, float32 in the kernel
            assert got[k] == pytest.approx(ref, rel=1e-7), k
        else:
            assert got[k] == ref, k


def test_level_commands_against_the_executed_reference(oracle_lib):
    """sample_velocity_command's curriculum branch (KAMMA.py:533-567) executed per arena. Level 1 has no draw: the only
    difference is math.sin in double vs the oracle's float32 polynomial. Levels 0 / 2 / else draw torch.normal(mean, std):
    the fixture holds the standard normal behind each draw (same seed through torch.randn), and mean + std z is formed
    in double there, in float32 here: a few ulp. Where std < 0 the reference's torch.normal raises; the build's value
    there is its own (docs/SPEC.md section 2.12)."""
    d = np.load(os.path.join(G, "refexec_level_commands.npz"))
    n = len(d["z"])
    ok = d["raised"] == 0
    c0, c2 = np.zeros(n, np.float32), np.zeros(n, np.float32)
    for i in range(n):
        a, b = C.c_float(), C.c_float()
        oracle_lib.orc_level_command(int(d["terrain_levels"][i]), float(d["fraction"][i]), int(d["episode_step_counter"][i]),
                                     float(d["z"][i]), float(d["dt"]), float(d["max_episode_length_s"]), C.byref(a), C.byref(b))
        c0[i], c2[i] = a.value, b.value
    lv = d["terrain_levels"]
    for level in range(5):
        assert (ok & (lv == level)).sum() >= 10, level
    np.testing.assert_allclose(c0[ok], d["commands_0"][ok], rtol=0, atol=6e-7)
    np.testing.assert_allclose(c2[ok], d["commands_2"][ok], rtol=0, atol=6e-7)
    m = ok & (lv == 0)      # level 0 returns (0.0, max(x_vel, 0.0)): the speed lands in commands[:, 2] (:541, :724-725)
    assert not c0[m].any() and (c2[m] > 0).any()
    m = ok & (lv != 0)
    assert not c2[m].any() and (d["commands_0"][m] >= 0).all()
    # the negative-sigma rows are exactly those the reference cannot evaluate
    sigma_neg = ((lv == 0) & (0.01 + 0.09 * d["fraction"] < 0)) | ((lv == 2) & (0.5 * d["fraction"] < 0))
    assert np.array_equal(sigma_neg, d["raised"].astype(bool))


def test_eval_ramp_against_the_executed_reference(oracle_lib):
    d = np.load(os.path.join(G, "refexec_eval_commands.npz"))  # KAMMA_eval.py:545-550
    got = np.array([oracle_lib.orc_eval_command(int(s), float(d["max_sim_steps"])) for s in d["sim_steps"]], np.float32)
    assert np.array_equal(got, d["commands"][:, 0]) and not d["commands"][:, 1].any()
    assert got.min() == 1.0 and got.max() == 2.0 and len(np.unique(got)) > 40


def test_gae_against_the_executed_reference():
    from oracle import binding as ob
    d = np.load(os.path.join(G, "refexec_gae.npz"))  # ppo.py:320-360, the nested compute_gae itself
    g, l = float(d["discount_factor"]), float(d["lambda_coefficient"])
    ret, adv = ob.gae(0, d["rewards"], d["terminated"], d["values"], d["last_values"], g, l, True)
    assert np.array_equal(ret, d["returns"])
    np.testing.assert_allclose(adv, d["advantages"], rtol=1e-5, atol=1e-6)  # mean / std reductions: summation order
    ret, raw = ob.gae(0, d["rewards"], d["terminated"], d["values"], d["last_values"], g, l, False)
    assert np.array_equal(raw + d["values"], d["returns"])                   # returns = advantages + values (:352)
    # PPOC.compute_advantages_and_returns (ppoc.py:305-336), the method itself executed: mode 1 of orc_gae / ois_gae, bit for bit
    ret2, adv2 = ob.gae(1, d["rewards"], d["terminated"], d["values"], d["last_values"], g, l, False)
    assert np.array_equal(adv2, d["ppoc_advantages"]) and np.array_equal(ret2, d["ppoc_returns"])


HS, VS = 0.02, 0.006


def _tile(n=200, m=None):
    return tg.SubTerrain("t", width=n, length=m or n, vertical_scale=VS, horizontal_scale=HS)


FAMILIES = {
    "flat": lambda r: tg.flat_terrain(_tile(50), height=0.3),
    "sloped": lambda r: tg.sloped_terrain(_tile(100), slope=0.25),
    "pyramid_sloped_pos": lambda r: tg.pyramid_sloped_terrain(_tile(200), slope=0.3, platform_size=1.0),
    "pyramid_sloped_neg": lambda r: tg.pyramid_sloped_terrain(_tile(200), slope=-0.2, platform_size=3.0),
    "stairs": lambda r: tg.stairs_terrain(_tile(100), step_width=0.3, step_height=0.05),
    "pyramid_stairs_up": lambda r: tg.pyramid_stairs_terrain(_tile(400), step_width=0.31, step_height=0.15, platform_size=3.0),
    "pyramid_stairs_down": lambda r: tg.pyramid_stairs_terrain(_tile(400), step_width=1.0, step_height=-0.047, platform_size=3.0),
    "wave": lambda r: tg.wave_terrain(_tile(100), num_waves=2, amplitude=0.2),
    "discrete_obstacles": lambda r: tg.discrete_obstacles_terrain(_tile(400), 0.15, 1.0, 2.0, 40, platform_size=3.0, rng=r),
    "random_uniform": lambda r: tg.random_uniform_terrain(_tile(100), -0.1, 0.1, step=0.05, downsampled_scale=0.2, rng=r),
    "stepping_stones": lambda r: tg.stepping_stones_terrain(_tile(200), 0.5, 0.1, 0.05, platform_size=1.0, rng=r),
    "stepping_stones_wide": lambda r: tg.stepping_stones_terrain(_tile(120, 200), 0.4, 0.08, 0.1, platform_size=1.0, rng=r),
}


def test_terrain_families_equal_the_executed_reference():
    """origin_terrain_utils.py:39-346 run here on the same tiles; the random families under the same numpy seed (the
    reference draws from numpy's global RandomState, terrain.py from RandomState(seed): the same stream, and the draws
    are made in the same order). Every cell equal."""
    d = np.load(os.path.join(G, "refexec_terrain.npz"))
    for k, (name, make) in enumerate(FAMILIES.items()):
        got = make(np.random.RandomState(100 + k)).height_field_raw
        assert got.dtype == np.int16 and got.shape == d[name].shape, name
        assert np.array_equal(got, d[name]), name


@pytest.mark.parametrize("tag,over,seed", [("gen_curriculum", {"curriculum": True}, 0), ("gen_random_s3", {}, 3), ("gen_random_s42", {}, 42),
                                           ("gen_random_small_s7", {"numLevels": 3, "numTerrains": 3, "mapLength": 8.0, "mapWidth": 8.0}, 7)])
def test_terrain_generator_equals_the_executed_reference(tag, over, seed):
    """origin_terrain_generator.py:37-183: the whole int16 field (sha256 over its bytes) and the tile origins of the
    curriculum layout and of the randomised one, numpy's global RNG seeded there, RandomState(seed) here."""
    d = np.load(os.path.join(G, "refexec_terrain.npz"))
    cfg = {"mapLength": 10.0, "mapWidth": 10.0, "numLevels": 6, "numTerrains": 2, "curriculum": False, **over}
    T = tg.Terrain(cfg, 4096, seed=seed)
    f = np.ascontiguousarray(T.height_field_raw)
    assert f.shape == tuple(d[tag + "_shape"]) and f.dtype == np.int16
    assert np.array_equal(f[::25, ::25], d[tag + "_every25"])
    assert hashlib.sha256(f.tobytes()).hexdigest() == str(d[tag + "_sha256"])
    assert np.array_equal(T.env_origins, d[tag + "_origins"])


VARIANT_FILES = ["KAMMA", "KAMMA_medium", "KAMMA_g11_g21", "KAMMA_g1d_g21", "KAMMA_curr1", "KAMMA_curr2", "KAMMA_eval",
                 "KA_DDPG_4D", "KA_DDPG_2D", "KA_DDPG_1D", "KA_DDPG_2D_eval", "KA_DDPG_1D_eval"]


def _leaves(d, path=""):
    if isinstance(d, dict):
        for k in sorted(d):
            yield from _leaves(d[k], path + "/" + k)
    else:
        yield path, d


def _set_leaf(cfg, path, value):
    keys = path.strip("/").split("/")
    for k in keys[:-1]:
        cfg = cfg.setdefault(k, {})
    cfg[keys[-1]] = value


def test_default_task_cfg_is_the_reference_literal_with_the_scripts_overrides():
    """default_task_cfg() against the TASK_CFG literal of KAMMA.py:29-128 (ast.literal_eval of the reference's file at
    fixture time, the scripts' argparse defaults written in): the digest over all shared leaves is the same, and the only
    leaves not carried are PhysX's GPU buffer sizes."""
    import json
    from options_in_simulation_amd import default_task_cfg
    d = np.load(os.path.join(G, "refexec_config.npz"))
    mine = dict(_leaves(default_task_cfg()))
    assert len(mine) == int(d["kamma_leaf_count"])
    assert hashlib.sha256(json.dumps(mine, sort_keys=True).encode()).hexdigest() == str(d["kamma_sha256"])
    dropped = [str(p) for p in d["kamma_dropped_paths"]]
    assert dropped and all(p.startswith("/task/sim/physx/gpu_") for p in dropped), dropped


@pytest.mark.parametrize("name", VARIANT_FILES)
def test_build_config_derives_what_update_config_derives(name):
    """update_config (KAMMA.py:231-275 and the same method of the 11 other env files) EXECUTED on each file's own TASK_CFG
    (the fixture holds the leaves in which it differs from KAMMA.py's, and the results): dt, max_episode_length =
    int(10 / 0.1 + 0.5), Kp, ranges, init state, sampler flags — against build_config(variant) on the same dict."""
    import json
    from options_in_simulation_amd import build_config, default_task_cfg
    d = np.load(os.path.join(G, "refexec_config.npz"))
    cfg = default_task_cfg()
    for path, v in json.loads(str(d[name + "__diff_json"])).items():
        assert v != "<absent>", path
        _set_leaf(cfg, path, v)
    cfg["task"]["env"]["numEnvs"] = json.loads(str(d["script_overrides_json"]))["numEnvs"]
    r = json.loads(str(d[name + "__derived_json"]))
    c = build_config(task_cfg=cfg, variant=name)
    assert c.num_envs == r["_num_envs"] == 1
    assert c.decimation == r["decimation"] and c.sim_dt == np.float32(r["sim_dt"])
    assert c.policy_dt == np.float32(r["dt"]) and c.max_episode_length == r["max_episode_length"] == 100
    assert c.episode_length_s == np.float32(r["max_episode_length_s"])
    assert c.Kp == np.float32(r["Kp"]) and c.action_scale == np.float32(r["action_scale"])
    assert c.terminal_reward == np.float32(r["termination"])
    assert [c.cmd_x_lo, c.cmd_x_hi] == [np.float32(v) for v in r["command_x_range"]]
    assert [c.cmd_y_lo, c.cmd_y_hi] == [np.float32(v) for v in r["command_y_range"]]
    assert [c.cmd_yaw_lo, c.cmd_yaw_hi] == [np.float32(v) for v in r["command_yaw_range"]]
    assert c.curriculum_threshold_high == np.float32(r["threshold_high"])
    init = r["base_init_state"]
    assert [c.init_pos[k] for k in range(3)] == [np.float32(v) for v in init[:3]]
    assert [c.init_quat[k] for k in range(4)] == [np.float32(v) for v in init[3:7]]
    assert [c.init_lin_vel[k] for k in range(3)] + [c.init_ang_vel[k] for k in range(3)] == [np.float32(v) for v in init[7:13]]
    assert [c.init_dof_vel[k] for k in range(4)] == [np.float32(v) for v in r["dof_init_state"][4:8]]
    sampler = (_abi.CMD_LEVEL if r["curriculum"] else _abi.CMD_EVAL if r["evalsampling"] else _abi.CMD_RAND if r["randsampling"]
               else _abi.CMD_BOX if r["boxsampling"] else _abi.CMD_GRID if r["gridsampling"] else None)
    assert c.cmd_sampler == sampler, (name, r)


RETURNS_SPEC = dict(variant="KAMMA_medium", n=24, steps=210, seed=13)


def _returns_run(env, to_numpy):
    sp = RETURNS_SPEC
    rng = np.random.default_rng(sp["seed"])
    rew, done = [], []
    for _ in range(sp["steps"]):
        env.step(rng.uniform(0, 1, (sp["n"], 4)).astype(np.float32))
        rew.append(to_numpy(env.buffer(_abi.BUF_REW)).copy()); done.append(to_numpy(env.buffer(_abi.BUF_DONE)).astype(np.uint8).copy())
    return np.array(rew), np.array(done), env.stats()


def _check_returns(rew, done, st):
    """the finished-episode statistics of the environment against Agent.record_transition (src/my_agents/base.py:270-341)
    EXECUTED on the same run (fixture): number of episodes, their total rewards (max / min / mean of the deque), lengths"""
    d = np.load(os.path.join(G, "refexec_returns.npz"))
    assert np.array_equal(rew, d["rewards"]) and np.array_equal(done, d["done"])  # the run the fixture was made from
    tr, tt = d["track_rewards"], d["track_timesteps"]
    assert st.episodes == len(tr) == 2 * RETURNS_SPEC["n"]
    assert st.return_max == np.float32(tr.max()) and st.return_min == np.float32(tr.min())        # torch float32 sums there
    assert st.return_sum / st.episodes == pytest.approx(d["td__Reward / Total reward (mean)"][-1], rel=1e-6)
    assert st.return_sum / st.episodes == pytest.approx(tr.mean(), rel=1e-6)
    assert st.length_sum / st.episodes == d["td__Episode / Total timesteps (mean)"][-1] == tt.mean() == 99.0
    var = st.return_sqsum / st.episodes - (st.return_sum / st.episodes) ** 2
    assert var == pytest.approx(tr.var(), rel=1e-5)
    # per-step instantaneous rewards: what the agent logs is torch.max / min / mean over the arenas of rew_buf
    assert np.array_equal(d["td__Reward / Instantaneous reward (max)"], rew.max(1).astype(np.float64))
    assert np.array_equal(d["td__Reward / Instantaneous reward (min)"], rew.min(1).astype(np.float64))


def test_episode_statistics_against_the_executed_agent_bookkeeping():
    from options_in_simulation_amd import build_config
    from oracle.binding import OracleEnv
    sp = RETURNS_SPEC
    env = OracleEnv(build_config(variant=sp["variant"], num_envs=sp["n"], seed=sp["seed"]))
    env.reset(sp["seed"])
    _check_returns(*_returns_run(env, np.asarray))


@pytest.mark.gpu
def test_episode_statistics_of_the_hip_path_against_the_executed_agent_bookkeeping():
    import torch
    from options_in_simulation_amd import build_config
    from options_in_simulation_amd.vecenv import NativeEnv
    sp = RETURNS_SPEC
    env = NativeEnv(build_config(variant=sp["variant"], num_envs=sp["n"], seed=sp["seed"]), "cuda:0")
    env.reset(sp["seed"])

    class Adapter:  # numpy actions in, as the oracle env takes them
        buffer, stats = env.buffer, env.stats

        @staticmethod
        def step(a):
            env.step(torch.from_numpy(a).cuda())
    _check_returns(*_returns_run(Adapter, lambda t: t.cpu().numpy()))


def test_grid_sampler_against_the_executed_reference(oracle_lib):
    """KAMMA_curr2.py:600-640 executed; the fixture holds the uniform and the standard normal behind every draw (same
    seed replayed), so the bimodal formula is compared exactly: float32 progress, means, z * 0.05 + mean, max(., 0)."""
    d = np.load(os.path.join(G, "refexec_grid_commands.npz"))
    lo, hi = (float(v) for v in d["x_range"])
    got = np.array([oracle_lib.orc_grid_command(int(s), float(d["max_sim_steps"]), lo, hi, float(u), float(z))
                    for s, u, z in zip(d["sim_steps"], d["u_side"], d["z"])], np.float32)
    assert np.array_equal(got, d["commands"][:, 0]) and not d["commands"][:, 1].any()
    late = d["sim_steps"] >= 500000                      # progress clamps at 1: the modes sit on the range ends
    assert (np.abs(got[late] - np.where(d["u_side"][late] < 0.5, lo, hi)) < 0.3).all()
    assert (d["u_side"] < 0.5).sum() > 20 and (d["u_side"] >= 0.5).sum() > 20
