#!/usr/bin/env python3
"""Golden vectors made by EXECUTING the reference's own code (build container only; needs /root/reference).

The reference's modules for this path cannot be imported whole — each one imports omni / pxr / gym / skrl at
module level (plain ModuleNotFoundError) — but several of the functions on and around the
hot path are pure torch / numpy / scipy / math over their arguments. This script reads the reference's files,
takes exactly those function / class definitions out of the syntax tree (ast), compiles them unchanged and runs
them on seeded inputs. No stand-in is supplied for any absent library: a function whose body needs one
(quat_apply, torch_rand_float, SimulationContext, skrl memories ...) is not used here and stays pinned by the
transcription in make_golden.py only. `self` is a plain namespace holding the tensors the method reads.

What is executed (file:line of the definitions in /root/reference/src):
  my_envs/*.py (12 files)     the TASK_CFG literal :29-128 (ast.literal_eval), TorqueDistributionTask.update_config :231
  my_envs/KAMMA.py            TorqueDistributionTask.calculate_metrics :767, .get_observations :825,
                              .sample_velocity_command :531 (curriculum branch)
  my_envs/KA_DDPG_4D.py       .calculate_metrics (single gamma_assist)
  my_envs/KAMMA_eval.py       .calculate_metrics (Tracking error sum), .sample_velocity_command (EvalSampling)
  my_envs/KAMMA_curr2.py      .sample_velocity_command (GridSampling :600-640)
  my_envs/KAMMA_g1d_g21.py    .get_observations (env-0 reward guard :838-842)
  my_agents/ppo.py            PPO._update.compute_gae :320 (nested function)
  my_agents/ppoc.py           PPOC.compute_advantages_and_returns :305
  my_agents/base.py           Agent.record_transition :270 (episode return / length bookkeeping)
  options_sim/cli/artifacts.py          _collect_timeseries :68, _write_csv :96 (over the build's event files)
  options_sim/cli/eval_artifacts.py     _get_series :62, _align_by_step :76, _moving_average :90
  my_utils/origin_terrain_utils.py      the generator families :39-346, SubTerrain :437
  my_utils/origin_terrain_generator.py  Terrain :37-183 (numpy's global RNG seeded here)

Outputs: tests/golden/refexec_*.npz (inputs and outputs only — data). Where make_golden.py transcribes the same
lines, this script also asserts that the transcription gives the same numbers on the same inputs.

    python tests/golden/make_golden_ref.py
"""
import ast
import hashlib
import math
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference/src"
F = torch.float32


def extract(relpath, *qualname, extra_globals=None):
    """the definition `qualname` (nested: class -> method -> inner function) of a reference file, compiled alone"""
    path = os.path.join(REF, relpath)
    node = ast.parse(open(path).read(), filename=path)
    for name in qualname:
        found = None
        for child in ast.walk(node):
            if child is not node and isinstance(child, (ast.FunctionDef, ast.ClassDef)) and child.name == name:
                found = child
                break
        if found is None:
            raise KeyError(f"{relpath}: {'.'.join(qualname)} not found")
        node = found
    node.decorator_list = []
    mod = ast.Module(body=[node], type_ignores=[])
    ast.fix_missing_locations(mod)
    ns = {"torch": torch, "np": np, "math": math}
    ns.update(extra_globals or {})
    exec(compile(mod, path, "exec"), ns)
    return ns[node.name], ns


SUM_KEYS = ["r1: Tracking error reward (squared errors)", "r2: Convergence reward (squared accelerations)",
            "r3: Torque penalty (sum of squared torques)", "Dense reward", "Sparse reward", "Guiding reward",
            "Observed reward", "Final reward"]


# ------------------------------------------------------------------ TASK_CFG / update_config ----
VARIANT_FILES = ["KAMMA", "KAMMA_medium", "KAMMA_g11_g21", "KAMMA_g1d_g21", "KAMMA_curr1", "KAMMA_curr2", "KAMMA_eval",
                 "KA_DDPG_4D", "KA_DDPG_2D", "KA_DDPG_1D", "KA_DDPG_2D_eval", "KA_DDPG_1D_eval"]
# what every train / eval script writes into TASK_CFG before it builds the task (argparse defaults,
# train/train_kamma_4d_nofifo_random_RLIL.py:62-68, 79-91)
SCRIPT_OVERRIDES = {"stiffness": 0.035, "damping": 0.005, "static_friction": 0.85, "dynamic_friction": 0.85,
                    "yaw_constant": 0.5, "linear_x": [1.0, 2.0], "numEnvs": 1}


def _leaves(d, path=""):
    if isinstance(d, dict):
        for k in sorted(d):
            yield from _leaves(d[k], path + "/" + k)
    else:
        yield path, d


def _apply_overrides(cfg):
    import copy
    run, o = copy.deepcopy(cfg), SCRIPT_OVERRIDES
    run["task"]["env"]["control"]["stiffness"], run["task"]["env"]["control"]["damping"] = o["stiffness"], o["damping"]
    run["task"]["sim"]["default_physics_material"]["static_friction"] = o["static_friction"]
    run["task"]["sim"]["default_physics_material"]["dynamic_friction"] = o["dynamic_friction"]
    run["task"]["env"]["randomCommandVelocityRanges"]["yaw_constant"] = o["yaw_constant"]
    run["task"]["env"]["randomCommandVelocityRanges"]["linear_x"] = o["linear_x"]
    return run


def config_cases():
    """the module-level TASK_CFG literal of every env file (ast.literal_eval) and what update_config (KAMMA.py:231-275)
    derives from it once the scripts' overrides are in. Stored: a digest of KAMMA.py's literal (with the overrides, without
    the keys default_task_cfg() does not carry, which are listed), the leaves in which each variant's literal differs from
    it, and update_config's results — not the literal itself."""
    import json
    from options_in_simulation_amd import default_task_cfg
    out, lits = {}, {}
    for name in VARIANT_FILES:
        rel = f"my_envs/{name}.py"
        tree = ast.parse(open(os.path.join(REF, rel)).read())
        node = next(n for n in tree.body if isinstance(n, ast.Assign) and getattr(n.targets[0], "id", None) == "TASK_CFG")
        lits[name] = ast.literal_eval(node.value)
    base = dict(_leaves(_apply_overrides(lits["KAMMA"])))
    mine = dict(_leaves(default_task_cfg()))
    dropped = sorted(set(base) - set(mine))
    kept = {k: v for k, v in base.items() if k in mine}
    out["kamma_leaf_count"] = np.int64(len(kept))
    out["kamma_dropped_paths"] = np.array(dropped)
    out["kamma_sha256"] = np.array(hashlib.sha256(json.dumps(kept, sort_keys=True).encode()).hexdigest())
    for name in VARIANT_FILES:
        lv = dict(_leaves(lits[name]))
        l0 = dict(_leaves(lits["KAMMA"]))
        diff = {k: lv.get(k, "<absent>") for k in sorted(set(lv) | set(l0)) if lv.get(k, "<absent>") != l0.get(k, "<absent>")}
        out[name + "__diff_json"] = np.array(json.dumps(diff, sort_keys=True))
        run = _apply_overrides(lits[name])
        run["task"]["env"]["numEnvs"] = SCRIPT_OVERRIDES["numEnvs"]
        fn, _ = extract(f"my_envs/{name}.py", "TorqueDistributionTask", "update_config")
        S = types.SimpleNamespace()
        fn(S, types.SimpleNamespace(config=run, task_config=run["task"]))
        derived = {k: getattr(S, k) for k in ("action_scale", "command_x_range", "command_y_range", "command_yaw_range", "yaw_constant",
                                              "base_init_state", "decimation", "sim_dt", "dt", "max_episode_length_s",
                                              "max_episode_length", "Kp", "Kd", "curriculum", "randsampling", "boxsampling",
                                              "gridsampling", "threshold_high", "_num_envs", "dof_init_state")}
        derived["termination"] = S.rew_scales["termination"]
        derived["evalsampling"] = bool(getattr(S, "evalsampling", False))
        out[name + "__derived_json"] = np.array(json.dumps(derived, sort_keys=True))
    out["script_overrides_json"] = np.array(json.dumps(SCRIPT_OVERRIDES, sort_keys=True))
    return out


# ------------------------------------------------------------------ calculate_metrics -----------
def metrics_case(relpath, seed, n=193, terminal=0.0, single_gamma=False, eval_sums=False):
    fn, _ = extract(relpath, "TorqueDistributionTask", "calculate_metrics")
    g = torch.Generator().manual_seed(seed)
    r = lambda *s, lo=-1.0, hi=1.0: (hi - lo) * torch.rand(*s, generator=g, dtype=F) + lo  # noqa: E731
    S = types.SimpleNamespace()
    S.device = "cpu"
    S.dt = 10 * 0.01
    S.v_forward_projected = r(n, lo=-0.5, hi=2.5)
    S.base_velocities = r(n, 6)
    S.previous_linear_velocity = r(n, 3, lo=-0.5, hi=2.5)
    S.previous_angular_velocity = r(n, 3)
    S.v_delta, S.omega_delta = r(n, lo=-1.0, hi=2.0), r(n, lo=-0.4, hi=0.4)
    S.v_delta[:7] = r(7, lo=-0.02, hi=0.02); S.omega_delta[:7] = r(7, lo=-0.02, hi=0.02)  # the sparse bonus
    S.wheel_torqs = torch.clip(r(n, 4, lo=-5.0, hi=5.0), -4.0, 4.0)
    gamma = torch.clamp(r(n, lo=-0.2, hi=1.2), 0.0, 1.0)
    if single_gamma:
        S.gamma_assist = gamma
    else:
        S.gamma_assist2 = gamma
    S.guiding_reward = -r(n, lo=0.0, hi=6.0)
    S.timeout_buf = (torch.rand(n, generator=g) < 0.3).long()
    S.reset_buf = ((torch.rand(n, generator=g) < 0.3) | S.timeout_buf.bool()).long()
    S.rew_scales = {"termination": terminal}
    keys = SUM_KEYS + (["Tracking error"] if eval_sums else [])
    S.episode_sums = {k: r(n, lo=-5.0, hi=5.0) for k in keys}
    inp = {"v_forward_projected": S.v_forward_projected.clone(), "omega_z": S.base_velocities[:, 5].clone(),
           "prev_lin": S.previous_linear_velocity[:, 0].clone(), "prev_ang": S.previous_angular_velocity[:, 2].clone(),
           "v_delta": S.v_delta.clone(), "omega_delta": S.omega_delta.clone(), "wheel_torqs": S.wheel_torqs.clone(),
           "gamma2": gamma.clone(), "guiding": S.guiding_reward.clone(), "reset": S.reset_buf.to(torch.uint8),
           "timeout": S.timeout_buf.to(torch.uint8), "terminal": torch.tensor(terminal, dtype=F),
           "sums_in": torch.stack([S.episode_sums[k] for k in keys], 1).clone()}
    ret = fn(S)
    assert ret is S.rew_buf
    out = {"rew": S.rew_buf, "lin_acc": S.linear_acc, "ang_acc": S.angular_acc,
           "prev_lin_out": S.previous_linear_velocity[:, 0], "prev_ang_out": S.previous_angular_velocity[:, 2],
           "sums_out": torch.stack([S.episode_sums[k] for k in keys], 1),
           "comp": torch.stack([S.comp_1, S.comp_2, S.comp_3, S.rdense, S.rsparse, S.robs, S.rguide, S.rfinal], 1)}
    d = {k: v.numpy() for k, v in {**inp, **out}.items()}
    d["source"] = np.array(relpath)
    return d


# ------------------------------------------------------------------ get_observations ------------
def observations_case(relpath, seed, n=5, tiny_rew0=False):
    fn, _ = extract(relpath, "TorqueDistributionTask", "get_observations")
    g = torch.Generator().manual_seed(seed)
    r = lambda *s, lo=-1.0, hi=1.0: (hi - lo) * torch.rand(*s, generator=g, dtype=F) + lo  # noqa: E731
    S = types.SimpleNamespace()
    S._robots = types.SimpleNamespace(name="robot_view")
    for k in ("desired_v", "desired_omega", "current_v", "current_omega", "v_delta", "omega_delta", "linear_acc", "angular_acc",
              "ac_left", "ac_right", "comp_1", "comp_2", "comp_3", "rdense", "rsparse", "robs", "rguide"):
        setattr(S, k, r(n, lo=-3.0, hi=3.0))
    S.linear_acc[1] = 2500.0  # beyond the wrapper's clip (rl_task.py:99): the task's own buffer is not clipped
    S.episode_count = torch.randint(0, 50, (n,), generator=g)
    S.torques, S.actions = r(n, 4, lo=-6.0, hi=6.0), r(n, 4, lo=0.0, hi=1.0)
    S.action_scale = 3.0
    S.gamma_assist2 = torch.clamp(r(n, lo=0.0, hi=1.0), 0.0, 1.0)
    S.rew_buf = r(n, lo=-9.0, hi=-0.1)
    if tiny_rew0:
        S.rew_buf[0] = 3.0e-13
    inp = {k: getattr(S, k).clone().numpy() for k in ("desired_v", "desired_omega", "current_v", "current_omega", "v_delta", "omega_delta",
                                                       "linear_acc", "angular_acc", "ac_left", "ac_right", "comp_1", "comp_2", "comp_3",
                                                       "rdense", "rsparse", "robs", "rguide", "torques", "actions", "gamma_assist2", "rew_buf")}
    inp["episode_count"] = S.episode_count.numpy().astype(np.int32)
    ret = fn(S)
    assert list(ret) == ["robot_view"] and ret["robot_view"]["obs_buf"] is S.obs_buf
    keys = list(S.observed_components)
    d = dict(inp)
    d.update(obs_buf=S.obs_buf.numpy(), rew_buf_out=S.rew_buf.numpy(), component_keys=np.array(keys),
             component_values=np.array([float(S.observed_components[k]) for k in keys], np.float64), source=np.array(relpath))
    return d


# ------------------------------------------------------------------ sample_velocity_command -----
def level_commands_case(seed=11, n=240):
    """curriculum branch (KAMMA.py:533-567) per arena as post_physics_step calls it (:719-725). torch.normal draws
    from torch's global generator: each call is preceded by torch.manual_seed(seed_i), and the standard normal
    behind the draw is recorded by replaying the same seed through torch.randn(1) (normal(mean, std) = mean + std z
    in double there, so a float32 replay agrees to an ulp or two: the test's tolerance). std < 0 makes the
    reference's torch.normal raise: recorded as such."""
    fn, _ = extract("my_envs/KAMMA.py", "TorqueDistributionTask", "sample_velocity_command")
    g = torch.Generator().manual_seed(seed)
    S = types.SimpleNamespace()
    S.curriculum, S.device = True, "cpu"
    S.terrain_levels = torch.randint(0, 5, (n,), generator=g)
    r1 = -torch.rand(n, generator=g, dtype=F) * 2.0
    r1[::3] = torch.rand(len(r1[::3]), generator=g, dtype=F) * 9.0
    S.episode_sums = {SUM_KEYS[0]: r1}
    S.threshold_high = 10.0
    S.fraction = r1 / S.threshold_high            # KAMMA.py:498
    S.dt = 10 * 0.01
    S.max_episode_length_s = 10.0
    counters = torch.randint(1, 500000, (n,), generator=g)
    cmd0, cmd2, z, raised = torch.zeros(n, dtype=F), torch.zeros(n, dtype=F), torch.zeros(n, dtype=F), np.zeros(n, np.uint8)
    for i in range(n):
        S.episode_step_counter = int(counters[i])
        torch.manual_seed(1000 + i)
        z[i] = torch.randn(1)[0]
        torch.manual_seed(1000 + i)
        try:
            a, b = fn(S, i)
        except RuntimeError:
            raised[i] = 1
            continue
        cmd0[i] = a   # KAMMA.py:724: the assignment rounds to float32
        cmd2[i] = b   # :725
    assert raised.sum() > 0 and raised.sum() < n // 2
    return {"terrain_levels": S.terrain_levels.numpy().astype(np.int32), "fraction": S.fraction.numpy(), "z": z.numpy(),
            "episode_step_counter": counters.numpy().astype(np.int32), "dt": np.float64(S.dt),
            "max_episode_length_s": np.float64(S.max_episode_length_s), "commands_0": cmd0.numpy(), "commands_2": cmd2.numpy(),
            "raised": raised}


def eval_commands_case(n=64):
    """KAMMA_eval.py:545-550: x = 1 + clamp(sim_steps / max_sim_steps, 0, 1) (no random draw)"""
    fn, _ = extract("my_envs/KAMMA_eval.py", "TorqueDistributionTask", "sample_velocity_command")
    S = types.SimpleNamespace()
    S.curriculum, S.evalsampling, S.device = False, True, "cpu"
    S.max_sim_steps = 50000.0
    S.sim_steps = torch.cat([torch.arange(0, 60000, 1000), torch.tensor([1, 49999, 50000, 50001])])[:n].long()
    out = np.zeros((len(S.sim_steps), 2), np.float32)
    for i in range(len(S.sim_steps)):
        a, b = fn(S, i)
        out[i] = [float(a), float(b)]
    return {"sim_steps": S.sim_steps.numpy().astype(np.int32), "max_sim_steps": np.float32(S.max_sim_steps), "commands": out}


def grid_commands_case(n=96):
    """KAMMA_curr2.py:600-640 (GridSampling), one arena (the branch's min(1.0, tensor) only works with one): the side
    pick torch.rand(1) < 0.5 and torch.normal(mean=Tensor, std=Tensor) = randn * std + mean in float32 — both replayed
    from the same seed, so the draw behind each output is in the fixture (u, z) and the formula can be checked exactly."""
    import warnings
    fn, _ = extract("my_envs/KAMMA_curr2.py", "TorqueDistributionTask", "sample_velocity_command")
    S = types.SimpleNamespace()
    S.curriculum, S.randsampling, S.boxsampling, S.gridsampling, S.device = False, False, False, True, "cpu"
    S.max_sim_steps = 500000.0
    S.command_x_range, S.command_yaw_range = [1.0, 2.0], [1.0, 1.1]
    steps = np.concatenate([np.linspace(0, 620000, n - 4).astype(np.int64), [1, 499999, 500000, 500001]])
    out = np.zeros((n, 2), np.float32)
    u, z = np.zeros(n, np.float32), np.zeros(n, np.float32)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")  # torch.tensor(tensor) copy-construct warning of :611
        for i, st in enumerate(steps):
            S.sim_steps = torch.tensor([int(st)])
            torch.manual_seed(5000 + i)
            u[i] = torch.rand(1)[0]; z[i] = torch.randn(1)[0]
            torch.manual_seed(5000 + i)
            a, b = fn(S, 0)
            out[i] = [float(a), float(b)]
    return {"sim_steps": steps.astype(np.int32), "max_sim_steps": np.float32(S.max_sim_steps), "x_range": np.array(S.command_x_range, np.float32),
            "u_side": u, "z": z, "commands": out}


# ------------------------------------------------------------------ GAE ------------------------
def gae_case():
    from tests.golden.make_golden import refmath_gae
    d = refmath_gae()
    fn, ns = extract("my_agents/ppo.py", "PPO", "_update", "compute_gae",
                     extra_globals={"last_values": torch.from_numpy(d["last_values"])})  # closure variable of _update (:362)
    returns, advantages = fn(torch.from_numpy(d["rewards"]), torch.from_numpy(d["terminated"]), torch.from_numpy(d["values"]), None,
                             discount_factor=float(d["discount_factor"]), lambda_coefficient=float(d["lambda_coefficient"]))
    assert np.array_equal(returns.numpy(), d["ppo_returns"]) and np.array_equal(advantages.numpy(), d["ppo_advantages"]), \
        "make_golden.refmath_gae differs from the reference's compute_gae"
    # PPOC.compute_advantages_and_returns (my_agents/ppoc.py:305-336), the whole method unchanged. It reads the rollout through
    # `self.memory.get_tensor_by_name` (skrl's memory: third-party; here an accessor over the fixture's tensors), passes the last
    # next state through the (empty = identity) preprocessors and asks `self.option_value` for its value: the fixture's
    # last_values. gamma / lam are the agent's Python floats.
    ppoc, _ = extract("my_agents/ppoc.py", "PPOC", "compute_advantages_and_returns")
    T = {"rewards": torch.from_numpy(d["rewards"]), "terminated": torch.from_numpy(d["terminated"]), "values": torch.from_numpy(d["values"]),
         "next_states": torch.zeros(d["rewards"].shape[0], d["rewards"].shape[1], 6), "current_option": torch.zeros_like(torch.from_numpy(d["rewards"]))}
    mem = types.SimpleNamespace(get_tensor_by_name=lambda name, idx=None: T[name] if idx is None else T[name][idx])
    agent = types.SimpleNamespace(memory=mem, gamma=float(d["discount_factor"]), lam=float(d["lambda_coefficient"]),
                                  _state_preprocessor=lambda x, **k: x, _value_preprocessor=lambda x, **k: x,
                                  option_value=lambda inputs, options: (torch.from_numpy(d["last_values"]), None, None))
    adv2, ret2 = ppoc(agent)
    assert np.array_equal(adv2.numpy(), d["ppoc_advantages"]) and np.array_equal(ret2.numpy(), d["ppoc_returns"]), \
        "make_golden.refmath_gae differs from the reference's PPOC.compute_advantages_and_returns"
    return {"rewards": d["rewards"], "values": d["values"], "last_values": d["last_values"], "terminated": d["terminated"],
            "discount_factor": d["discount_factor"], "lambda_coefficient": d["lambda_coefficient"],
            "returns": returns.numpy(), "advantages": advantages.numpy(),
            "ppoc_advantages": adv2.numpy(), "ppoc_returns": ret2.numpy()}


# ------------------------------------------------------------------ terrain ----------------------
def terrain_namespace():
    from numpy.random import choice
    from scipy import interpolate
    names = ["random_uniform_terrain", "flat_terrain", "sloped_terrain", "pyramid_sloped_terrain", "discrete_obstacles_terrain",
             "wave_terrain", "stairs_terrain", "pyramid_stairs_terrain", "stepping_stones_terrain", "convert_heightfield_to_trimesh",
             "SubTerrain"]
    ns = {"np": np, "choice": choice, "interpolate": interpolate, "sqrt": math.sqrt}
    for nm in names:
        obj, _ = extract("my_utils/origin_terrain_utils.py", nm, extra_globals=ns)
        ns[nm] = obj
    T, _ = extract("my_utils/origin_terrain_generator.py", "Terrain", extra_globals=ns)
    ns["Terrain"] = T
    return ns


def terrain_cases():
    ns = terrain_namespace()
    HS, VS = 0.02, 0.006
    tile = lambda n=200, m=None: ns["SubTerrain"]("t", width=n, length=m or n, vertical_scale=VS, horizontal_scale=HS)  # noqa: E731
    out = {}
    fams = {
        "flat": lambda: ns["flat_terrain"](tile(50), height=0.3),
        "sloped": lambda: ns["sloped_terrain"](tile(100), slope=0.25),
        "pyramid_sloped_pos": lambda: ns["pyramid_sloped_terrain"](tile(200), slope=0.3, platform_size=1.0),
        "pyramid_sloped_neg": lambda: ns["pyramid_sloped_terrain"](tile(200), slope=-0.2, platform_size=3.0),
        "stairs": lambda: ns["stairs_terrain"](tile(100), step_width=0.3, step_height=0.05),
        "pyramid_stairs_up": lambda: ns["pyramid_stairs_terrain"](tile(400), step_width=0.31, step_height=0.15, platform_size=3.0),
        "pyramid_stairs_down": lambda: ns["pyramid_stairs_terrain"](tile(400), step_width=1.0, step_height=-0.047, platform_size=3.0),
        "wave": lambda: ns["wave_terrain"](tile(100), num_waves=2, amplitude=0.2),
        "discrete_obstacles": lambda: ns["discrete_obstacles_terrain"](tile(400), 0.15, 1.0, 2.0, 40, platform_size=3.0),
        "random_uniform": lambda: ns["random_uniform_terrain"](tile(100), -0.1, 0.1, step=0.05, downsampled_scale=0.2),
        "stepping_stones": lambda: ns["stepping_stones_terrain"](tile(200), 0.5, 0.1, 0.05, platform_size=1.0),
        "stepping_stones_wide": lambda: ns["stepping_stones_terrain"](tile(120, 200), 0.4, 0.08, 0.1, platform_size=1.0),
    }
    for k, (name, make) in enumerate(fams.items()):
        np.random.seed(100 + k)
        t = make()
        out[name] = t.height_field_raw.copy()
        assert out[name].dtype == np.int16
    # the whole generator: curriculum layout (deterministic) and the randomised one under two seeds
    base = {"mapLength": 10.0, "mapWidth": 10.0, "numLevels": 6, "numTerrains": 2, "slopeTreshold": 0.5, "terrainProportions": [0.1, 0.5, 0.4]}
    for tag, cfg, seed in (("gen_curriculum", {**base, "curriculum": True}, 0), ("gen_random_s3", {**base, "curriculum": False}, 3),
                           ("gen_random_s42", {**base, "curriculum": False}, 42),
                           ("gen_random_small_s7", {**base, "curriculum": False, "numLevels": 3, "numTerrains": 3, "mapLength": 8.0, "mapWidth": 8.0}, 7)):
        np.random.seed(seed)
        T = ns["Terrain"](cfg, 4096)
        f = np.ascontiguousarray(T.height_field_raw)
        out[tag + "_shape"] = np.array(f.shape, np.int32)
        out[tag + "_sha256"] = np.array(hashlib.sha256(f.tobytes()).hexdigest())
        out[tag + "_origins"] = T.env_origins.copy()
        out[tag + "_every25"] = f[::25, ::25].copy()
        out[tag + "_tri"] = np.array([T.vertices.shape[0], T.triangles.shape[0]], np.int64)
    return out


# ------------------------------------------------------------------ artefact CLIs ---------------
def artifacts_case(seed=41):
    """src/options_sim/cli/artifacts.py _collect_timeseries :68 + _write_csv :96 and eval_artifacts.py _get_series :62,
    _align_by_step :76, _moving_average :90, run over an event file written by options_in_simulation_amd.tracking
    (TensorBoard's EventAccumulator is absent: the accumulator handed to the reference's functions is the build's own
    reader behind the two calls they make, Scalars(tag) -> events with .step / .value, KeyError for an unknown tag)."""
    import collections
    import tempfile
    from options_in_simulation_amd import tracking
    rng = np.random.default_rng(seed)
    steps = [800 * (i + 1) for i in range(12)]
    cols = ["v_delta", "desired_v", "current_v", "omega_delta", "desired_omega", "current_omega"]
    values = {c: rng.normal(size=len(steps)) * (1e-5 if "omega" in c else 1.0) for c in cols}
    te, dv = np.abs(rng.normal(size=len(steps))) * 0.4, np.linspace(1.0, 2.0, len(steps)) + rng.normal(size=len(steps)) * 1e-3
    Ev = collections.namedtuple("Ev", "wall_time step value")

    class EA:
        def __init__(self, path):
            self.s = collections.defaultdict(list)
            for step, wall, scalars in tracking.read_events(path):
                for tag, v in scalars.items():
                    self.s[tag].append(Ev(wall, step, v))

        def Scalars(self, tag):
            if tag not in self.s:
                raise KeyError(tag)
            return self.s[tag]

    tag_node = next(n for n in ast.parse(open(os.path.join(REF, "options_sim/cli/artifacts.py")).read()).body
                    if isinstance(n, ast.Assign) and getattr(n.targets[0], "id", None) == "TAG_CANDIDATES")
    import csv as _csv
    from pathlib import Path
    g = {"TAG_CANDIDATES": ast.literal_eval(tag_node.value), "csv": _csv, "Path": Path, "EventAccumulator": object}
    collect, _ = extract("options_sim/cli/artifacts.py", "_collect_timeseries", extra_globals=g)
    write_csv, _ = extract("options_sim/cli/artifacts.py", "_write_csv", extra_globals=g)
    cand = {}
    for nm in ("TRACKING_ERR_CANDS", "DESIRED_VEL_CANDS"):
        node = next(n for n in ast.parse(open(os.path.join(REF, "options_sim/cli/eval_artifacts.py")).read()).body
                    if isinstance(n, ast.Assign) and getattr(n.targets[0], "id", None) == nm)
        cand[nm] = ast.literal_eval(node.value)
    from typing import Dict, List, Optional, Tuple
    g2 = {"List": List, "Tuple": Tuple, "Dict": Dict, "Optional": Optional, "EventAccumulator": object}
    get_series, _ = extract("options_sim/cli/eval_artifacts.py", "_get_series", extra_globals=g2)
    align, _ = extract("options_sim/cli/eval_artifacts.py", "_align_by_step", extra_globals=g2)
    mavg, _ = extract("options_sim/cli/eval_artifacts.py", "_moving_average", extra_globals=g2)
    with tempfile.TemporaryDirectory() as tmp:
        w = tracking.EventFileWriter(tmp)
        for i, st in enumerate(steps):
            sc = {f"Reward_comp_env0 / {c}": float(values[c][i]) for c in cols if not (c == "current_v" and i == 4)}  # one gap -> NaN
            if i != 7:                                                           # an unmatched step on the eval side
                sc["Info / rew_Tracking error"] = float(te[i])
            sc["Info / rew_Desired velocity"] = float(dv[i])
            w.add_scalars(sc, st)
        w.close()
        path = [os.path.join(tmp, f) for f in os.listdir(tmp) if "tfevents" in f][0]
        ea = EA(path)
        st_out, data, matched = collect(ea)
        csv_path = write_csv(Path(tmp), st_out, data)
        csv_text = open(csv_path).read()
        te_tag, te_series = get_series(ea, cand["TRACKING_ERR_CANDS"])
        dv_tag, dv_series = get_series(ea, cand["DESIRED_VEL_CANDS"])
        xs, ys = align(dv_series, te_series)
        ys_s = mavg(list(ys), 4)
    assert matched["v_delta"] == "Reward_comp_env0 / v_delta" and te_tag == "Info / rew_Tracking error"
    out = {"steps": np.array(steps, np.int64), "tracking_error": te, "desired_velocity": dv, "timeseries_csv": np.array(csv_text),
           "speed": np.array(xs, np.float64), "tracking_error_aligned": np.array(ys, np.float64),
           "tracking_error_smooth4": np.array(ys_s, np.float64)}
    out.update({"value_" + c: values[c] for c in cols})
    return out


# ------------------------------------------------------------------ episode / return bookkeeping --
RETURNS_SPEC = dict(variant="KAMMA_medium", n=24, steps=210, seed=13, falls={})


def returns_case():
    """Agent.record_transition (src/my_agents/base.py:270-341): cumulative reward and length per arena, pushed to the
    100-deep deques when an episode ends, max / min / mean per step — executed on the rewards and done flags of an
    oracle run (RETURNS_SPEC: two synchronous timeouts = 48 finished episodes, so the deques do not wrap). Only
    synchronous endings are used: with rewards shaped [N, 1] the method's `(terminated + truncated).nonzero()` yields
    [i, 0] pairs and `self._cumulative_rewards[finished_episodes] = 0` then also clears row 0 whenever ANY arena finishes
    (inherited from skrl, invisible with the one arena the reference runs) — a quirk of the agent-side log, not of the
    environment, which the device-side statistics (ois_stats_fetch) do not imitate."""
    import collections
    from options_in_simulation_amd import _abi, build_config
    from oracle.binding import OracleEnv
    fn, _ = extract("my_agents/base.py", "Agent", "record_transition", extra_globals={"Any": object})
    sp = RETURNS_SPEC
    env = OracleEnv(build_config(variant=sp["variant"], num_envs=sp["n"], seed=sp["seed"]))
    env.reset(sp["seed"])
    A = types.SimpleNamespace(write_interval=800, _cumulative_rewards=None, _cumulative_timesteps=None,
                              _track_rewards=collections.deque(maxlen=100), _track_timesteps=collections.deque(maxlen=100),
                              tracking_data=collections.defaultdict(list))
    rng = np.random.default_rng(sp["seed"])
    rew, done = [], []
    for t in range(sp["steps"]):
        if t in sp["falls"]:
            env.buffer(_abi.BUF_QUAT)[sp["falls"][t]] = np.array([0.0, 1.0, 0.0, 0.0], np.float32)
        env.step(rng.uniform(0, 1, (sp["n"], 4)).astype(np.float32))
        r = torch.from_numpy(env.rew.copy()).view(-1, 1)
        d = torch.from_numpy(env.done.astype(bool).copy()).view(-1, 1)
        fn(A, None, None, r, None, d, torch.zeros_like(d), {}, t, sp["steps"])   # truncated is all zeros (wrapper)
        rew.append(r[:, 0].numpy().copy()); done.append(d[:, 0].numpy().copy())
    st = env.stats()
    out = {"rewards": np.array(rew), "done": np.array(done).astype(np.uint8),
           "track_rewards": np.array(A._track_rewards, np.float64), "track_timesteps": np.array(A._track_timesteps, np.int64)}
    for k, v in A.tracking_data.items():
        out["td__" + k] = np.array(v, np.float64)
    out["oracle_stats"] = np.array([st.episodes, st.return_sum, st.return_sqsum, st.return_min, st.return_max, st.length_sum], np.float64)
    assert len(A._track_rewards) == 2 * sp["n"] < 100 and set(A._track_timesteps) == {99}
    return out


def main():
    from tests.golden import make_golden as mg
    save = lambda name, d: np.savez_compressed(os.path.join(HERE, name), **d)  # noqa: E731
    cases = {"kamma_t0": metrics_case("my_envs/KAMMA.py", 21, terminal=0.0), "kamma_t15": metrics_case("my_envs/KAMMA.py", 22, terminal=1.5),
             "ddpg4d": metrics_case("my_envs/KA_DDPG_4D.py", 23, single_gamma=True),
             "eval": metrics_case("my_envs/KAMMA_eval.py", 24, eval_sums=True)}
    save("refexec_metrics.npz", {f"{c}__{k}": v for c, d in cases.items() for k, v in d.items()})
    obs = {"kamma": observations_case("my_envs/KAMMA.py", 31), "g1d": observations_case("my_envs/KAMMA_g1d_g21.py", 32, tiny_rew0=True),
           "g1d_normal": observations_case("my_envs/KAMMA_g1d_g21.py", 33)}
    save("refexec_observations.npz", {f"{c}__{k}": v for c, d in obs.items() for k, v in d.items()})
    lc = level_commands_case()
    save("refexec_level_commands.npz", lc)
    save("refexec_eval_commands.npz", eval_commands_case())
    save("refexec_grid_commands.npz", grid_commands_case())
    save("refexec_gae.npz", gae_case())
    save("refexec_terrain.npz", terrain_cases())
    save("refexec_config.npz", config_cases())
    save("refexec_artifacts.npz", artifacts_case())
    save("refexec_returns.npz", returns_case())
    # the transcription of make_golden.py against the executed reference, same inputs
    import tests.golden.make_golden  # noqa: F401
    d = mg.refmath_curriculum()
    print("wrote refexec_*.npz;", {k: int(v["rew"].size) for k, v in cases.items()}, "level-command rows", len(lc["z"]),
          "of which torch.normal raised", int(lc["raised"].sum()), "| transcription rows", len(d["z"]))


if __name__ == "__main__":
    main()
