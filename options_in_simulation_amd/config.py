"""Config surface of the skid-steer torque-control task.

Three layers, as in the reference (SURVEY.md §5 "Config / flags"):
  1. a nested ``TASK_CFG`` dict with the keys of src/my_envs/KAMMA.py:29-128 — existing
     train scripts patch exactly these keys (train_kamma_4d_nofifo_random_RLIL.py:63-91);
  2. a *variant* preset = the switches that distinguish the 12 copy-paste env files
     (SURVEY.md §2.1);
  3. the flat ``ois_config`` C struct (include/ois.h) handed to the native library.
"""
from __future__ import annotations

import copy
import ctypes as C
import math
from typing import Any, Dict, Optional

import numpy as np

from . import _abi

# --------------------------------------------------------------------------------------
# 1. TASK_CFG — same keys / nesting / values as KAMMA.py:29-128. Only the keys read by
#    update_config (KAMMA.py:231-275) and rl_task.py:99-103 influence the environment.
# --------------------------------------------------------------------------------------
_PHYSX = dict(worker_thread_count=4, solver_type=0, use_gpu=True,
              solver_position_iteration_count=4, solver_velocity_iteration_count=4,
              contact_offset=0.01, rest_offset=0.0, bounce_threshold_velocity=0.2,
              friction_offset_threshold=0.04, friction_correlation_distance=0.025,
              enable_sleeping=True, enable_stabilization=True, max_depenetration_velocity=100.0)

TASK_CFG: Dict[str, Any] = {
    "test": False, "device_id": 0, "headless": True, "sim_device": "gpu",
    "enable_livestream": True, "warp": False, "seed": 42,
    "task": {
        "name": "TorqueDistributionTask", "physics_engine": "physx",
        "env": {
            "numEnvs": 16, "envSpacing": 3.0, "episodeLength": 500, "enableDebugVis": False,
            "clipObservations": 1000.0, "controlFrequencyInv": 10,
            "baseInitState": {"pos": [0.0, 0.0, 0.1], "rot": [1.0, 0.0, 0.0, 0.0],
                              "vLinear": [0.0, 0.0, 0.0], "vAngular": [0.0, 0.0, 0.0]},
            "dofInitTorques": [0.0, 0.0, 0.0, 0.0],
            "dofInitVelocities": [0.0, 0.0, 0.0, 0.0],
            "terrain": {"staticFriction": 0.85, "dynamicFriction": 0.85, "restitution": 0.0,
                        "curriculum": False, "RandSampling": True, "BoxSampling": False,
                        "GridSampling": False, "maxInitMapLevel": 0, "mapLength": 10.0,
                        "mapWidth": 10.0, "numLevels": 6, "numTerrains": 2, "slopeTreshold": 0.5},
            "TerrainType": "double room",
            "learn": {"heightMeasurementScale": 1.0, "terminalReward": 0.0, "episodeLength_s": 10.0},
            "randomCommandVelocityRanges": {"linear_x": [1.5, 1.5], "linear_y": [-0.5, 0.5],
                                            "yaw": [1.0, 1.1], "yaw_constant": 0.5},
            "control": {"decimation": 10, "stiffness": 1.0, "damping": 0.005, "actionScale": 3.0,
                        "wheel_radius": 0.1175},
        },
        "sim": {
            "dt": 0.01, "use_gpu_pipeline": True, "gravity": [0.0, 0.0, -9.81],
            "add_ground_plane": True, "add_distant_light": True, "use_flatcache": True,
            "enable_scene_query_support": False, "enable_cameras": False,
            "default_physics_material": {"static_friction": 1.0, "dynamic_friction": 1.0,
                                         "restitution": 0.0},
            "physx": _PHYSX,
            "robot": {"override_usd_defaults": False, "fixed_base": False,
                      "enable_self_collisions": False, "enable_gyroscopic_forces": False,
                      "solver_position_iteration_count": 4, "solver_velocity_iteration_count": 4,
                      "sleep_threshold": 0.005, "stabilization_threshold": 0.001, "density": -1,
                      "max_depenetration_velocity": 100.0, "contact_offset": 0.005,
                      "rest_offset": 0.0},
        },
    },
}


def default_task_cfg(script_overrides: bool = True) -> Dict[str, Any]:
    """A fresh TASK_CFG. ``script_overrides`` applies what every train script patches in
    (train_kamma_4d_nofifo_random_RLIL.py:63-91): stiffness 0.035, damping 0.005, friction
    0.85/0.85, linear_x [1, 2], numEnvs 1 is left to the caller."""
    cfg = copy.deepcopy(TASK_CFG)
    if script_overrides:
        env = cfg["task"]["env"]
        env["control"]["stiffness"] = 0.035
        env["control"]["damping"] = 0.005
        env["terrain"]["staticFriction"] = 0.85
        env["terrain"]["dynamicFriction"] = 0.85
        mat = cfg["task"]["sim"].setdefault("default_physics_material", {})
        mat["static_friction"] = 0.85    # train_kamma_4d_nofifo_random_RLIL.py:88-89 (the key the scripts patch)
        mat["dynamic_friction"] = 0.85
        env["randomCommandVelocityRanges"]["linear_x"] = [1.0, 2.0]
    return cfg


# --------------------------------------------------------------------------------------
# 2. Variants. Each entry: the fields of ois_config that differ.
# --------------------------------------------------------------------------------------
A = _abi
VARIANTS: Dict[str, Dict[str, Any]] = {
    # KAMMA.py:179, 627-635, 656
    "KAMMA": dict(action_dim=4, action_low=0.0, mix_mode=A.MIX_SWITCH, max_sim_steps=400000.0),
    # KAMMA_medium.py:178-179 (the variant the shipped run used)
    "KAMMA_medium": dict(action_dim=4, action_low=0.0, mix_mode=A.MIX_SWITCH, max_sim_steps=500000.0),
    # KAMMA_g11_g21.py:627-628: both schedule lines commented -> gammas stay at their init 1
    "KAMMA_g11_g21": dict(action_dim=4, action_low=0.0, mix_mode=A.MIX_SWITCH, max_sim_steps=100000.0,
                          gamma1_mode=A.GAMMA_CONST, gamma1_const=1.0,
                          gamma2_mode=A.GAMMA_CONST, gamma2_const=1.0),
    # KAMMA_g1d_g21.py:628: gamma2 line commented; :838-842: rew_buf[0] = 1e-12 when it is (nearly) zero
    "KAMMA_g1d_g21": dict(action_dim=4, action_low=0.0, mix_mode=A.MIX_SWITCH, max_sim_steps=500000.0,
                          gamma2_mode=A.GAMMA_CONST, gamma2_const=1.0, env0_reward_guard=1),
    # KAMMA_curr1.py:56-57, 179, 577-598
    "KAMMA_curr1": dict(action_dim=4, action_low=0.0, mix_mode=A.MIX_SWITCH, max_sim_steps=500000.0,
                        cmd_sampler=A.CMD_BOX),
    # KAMMA_curr2.py:57-58, 179, 602-640
    "KAMMA_curr2": dict(action_dim=4, action_low=0.0, mix_mode=A.MIX_SWITCH, max_sim_steps=500000.0,
                        cmd_sampler=A.CMD_GRID),
    # KAMMA_eval.py:180, 193-194, 545-550, 646-647, 677-685
    "KAMMA_eval": dict(action_dim=4, action_low=0.0, mix_mode=A.MIX_SWITCH, max_sim_steps=50000.0,
                       gamma1_mode=A.GAMMA_CONST, gamma1_const=0.0,
                       gamma2_mode=A.GAMMA_CONST, gamma2_const=0.0,
                       cmd_sampler=A.CMD_EVAL, eval_sums=1),
    # KA_DDPG_4D.py:165, 179, 624-641
    "KA_DDPG_4D": dict(action_dim=4, action_low=-1.0, mix_mode=A.MIX_BLEND, max_sim_steps=700000.0,
                       guiding_mode=A.GUIDE_POLICY),
    # KA_DDPG_2D.py:157, 623-627 (gamma update commented -> stays at init 1)
    "KA_DDPG_2D": dict(action_dim=2, action_low=-1.0, mix_mode=A.MIX_BLEND, max_sim_steps=700000.0,
                       guiding_mode=A.GUIDE_POLICY,
                       gamma1_mode=A.GAMMA_CONST, gamma1_const=1.0,
                       gamma2_mode=A.GAMMA_CONST, gamma2_const=1.0),
    # KA_DDPG_1D.py:157, 600-603, 629 (gamma update commented -> stays at init 1)
    "KA_DDPG_1D": dict(action_dim=1, action_low=-1.0, mix_mode=A.MIX_BLEND, max_sim_steps=700000.0,
                       guiding_mode=A.GUIDE_POLICY,
                       gamma1_mode=A.GAMMA_CONST, gamma1_const=1.0,
                       gamma2_mode=A.GAMMA_CONST, gamma2_const=1.0),
    # KA_DDPG_2D_eval.py:180, 193, 542-548, 645, 685-692, 857 (gamma init 0 and never updated)
    "KA_DDPG_2D_eval": dict(action_dim=2, action_low=-1.0, mix_mode=A.MIX_BLEND, max_sim_steps=50000.0,
                            guiding_mode=A.GUIDE_POLICY, cmd_sampler=A.CMD_EVAL, eval_sums=1,
                            gamma1_mode=A.GAMMA_CONST, gamma1_const=0.0,
                            gamma2_mode=A.GAMMA_CONST, gamma2_const=0.0),
    # KA_DDPG_1D_eval.py:180, 193, 542-548, 647, 685-692, 857
    "KA_DDPG_1D_eval": dict(action_dim=1, action_low=-1.0, mix_mode=A.MIX_BLEND, max_sim_steps=50000.0,
                            guiding_mode=A.GUIDE_POLICY, cmd_sampler=A.CMD_EVAL, eval_sums=1,
                            gamma1_mode=A.GAMMA_CONST, gamma1_const=0.0,
                            gamma2_mode=A.GAMMA_CONST, gamma2_const=0.0),
}

# --------------------------------------------------------------------------------------
# Robot parameters. The only readable physical parameter source in the reference is
# src/my_assets/urdf/origin_sym.urdf (the USD is a binary crate); line numbers cite it.
# --------------------------------------------------------------------------------------
ROBOT = dict(
    chassis_mass=21.0,                 # urdf:11
    chassis_inertia=(0.4, 0.77, 1.01), # urdf:12, about the base-link origin (no inertial origin)
    wheel_mass=1.0,                    # urdf:34, 54, 75, 95
    wheel_inertia=0.01,                # urdf:35 (isotropic)
    wheel_radius=0.1175,               # urdf:40; KAMMA.py:79
    wheel_pos=(0.205, 0.25, 0.056),    # urdf:115-139 (|x|, |y|, z); left = +y, front = +x
    max_angular_velocity=64.0,         # origin_v18.py:70 (64/pi*180 deg/s)
    # Wheel joint resistance. The USD's joint drive is a binary crate; what the text sources give is the
    # URDF's joint velocity limit, <limit velocity="10"/> (urdf:117, 125, 133, 141) = 1.175 m/s at the
    # rim, inside the commanded 1-2 m/s. Model (docs/SPEC.md 3.3): T = tau - c1 w - c2 (w - clamp(w, +-10)).
    # c1, c2 are CALIBRATED on the shipped training log (tools/calibrate_physics.py --fit ->
    # tests/golden/ref_stats.json): expert-only runs then give v_delta 0.164 (log 0.1645), the
    # log's v_delta(desired_v) line to 0.03 m/s, and the README figure's U-shaped controller curve
    # (0.45 / 0.15 / 0.29 against 0.42 / 0.15 / 0.33). Calibration, not parity.
    wheel_damping=0.0,
    wheel_limit_velocity=10.0,
    wheel_limit_damping=0.6,
    baumgarte=0.2,
)


def composite_body(robot: Dict[str, Any] = ROBOT):
    """Mass, COM height and principal inertia of chassis + 4 wheels about the composite COM.
    The wheel's own spin inertia about the axle (body y) is excluded: it belongs to the
    wheel-spin degree of freedom (docs/SPEC.md §3.1)."""
    mc, mw = robot["chassis_mass"], robot["wheel_mass"]
    wx, wy, wz = robot["wheel_pos"]
    iw = robot["wheel_inertia"]
    m = mc + 4.0 * mw
    com = 4.0 * mw * wz / m
    ixx, iyy, izz = robot["chassis_inertia"]
    dz = wz - com
    ixx = ixx + mc * com * com + 4.0 * (mw * (wy * wy + dz * dz) + iw)
    iyy = iyy + mc * com * com + 4.0 * (mw * (wx * wx + dz * dz))
    izz = izz + 4.0 * (mw * (wx * wx + wy * wy) + iw)
    return m, com, (ixx, iyy, izz)


def build_config(task_cfg: Optional[Dict[str, Any]] = None, variant: str = "KAMMA_medium",
                 num_envs: Optional[int] = None, arena_offset: int = 0, seed: Optional[int] = None,
                 extra_substeps: Optional[int] = None, **overrides) -> _abi.OisConfig:
    """TASK_CFG dict + variant -> ois_config. Mirrors update_config (KAMMA.py:231-275) and
    the constructor constants (KAMMA.py:174-179, 187)."""
    if variant not in VARIANTS:
        raise ValueError(f"unknown variant {variant!r}; choose from {sorted(VARIANTS)}")
    cfg = task_cfg if task_cfg is not None else default_task_cfg()
    tcfg = cfg["task"] if "task" in cfg else cfg
    env, sim = tcfg["env"], tcfg["sim"]
    var = dict(VARIANTS[variant])
    var.pop("action_low", None)

    c = _abi.OisConfig()
    c.struct_bytes = C.sizeof(_abi.OisConfig)
    c.num_envs = int(num_envs if num_envs is not None else env["numEnvs"])   # KAMMA.py:272
    c.arena_offset = int(arena_offset)
    c.seed = int(seed if seed is not None else cfg.get("seed", 42)) & 0xFFFFFFFFFFFFFFFF
    # variant defaults
    c.action_dim, c.mix_mode = 4, _abi.MIX_SWITCH
    c.gamma1_mode = c.gamma2_mode = _abi.GAMMA_SCHEDULE
    c.gamma1_const = c.gamma2_const = 1.0
    c.guiding_mode, c.cmd_sampler, c.eval_sums = _abi.GUIDE_EXECUTED, _abi.CMD_RAND, 0
    c.max_sim_steps = 400000.0
    terrain = env.get("terrain", {})
    if terrain.get("BoxSampling") and not terrain.get("RandSampling", True):
        c.cmd_sampler = _abi.CMD_BOX
    if terrain.get("GridSampling") and not terrain.get("RandSampling", True) and not terrain.get("BoxSampling"):
        c.cmd_sampler = _abi.CMD_GRID
    if terrain.get("EvalSampling"):
        c.cmd_sampler = _abi.CMD_EVAL
    for k, v in var.items():
        setattr(c, k, v)
    if terrain.get("curriculum"):  # sample_velocity_command tests self.curriculum first (KAMMA.py:533)
        if c.cmd_sampler == _abi.CMD_EVAL:
            # the *_eval files test self.evalsampling BEFORE self.curriculum (KAMMA_eval.py:544), so there the per-step
            # re-draw of the curriculum (:719-725) yields the eval ramp: a combination no script uses and the kernels
            # do not implement
            raise ValueError("terrain.curriculum with EvalSampling is not supported")
        c.cmd_sampler = _abi.CMD_LEVEL
    c.curriculum_threshold_high = 10.0                            # KAMMA.py:269

    ctrl, learn = env["control"], env["learn"]
    c.sim_dt = float(sim["dt"])                                   # KAMMA.py:259
    c.decimation = int(ctrl["decimation"])                        # KAMMA.py:258
    c.extra_substeps = int(extra_substeps if extra_substeps is not None
                           else env.get("controlFrequencyInv", 10))  # KAMMA.py:43, rl_task.py:103
    dt = c.decimation * float(sim["dt"])                          # KAMMA.py:260 (Python double)
    c.policy_dt = dt
    c.episode_length_s = float(learn["episodeLength_s"])          # KAMMA.py:261
    c.max_episode_length = int(float(learn["episodeLength_s"]) / dt + 0.5)  # KAMMA.py:262
    c.Kp = float(ctrl["stiffness"])                               # KAMMA.py:263
    c.Kp_omega = 0.665                                            # KAMMA.py:613
    c.vehicle_mass, c.vehicle_inertia = 25.0, 1.05                # KAMMA.py:174-175
    c.action_scale = float(ctrl["actionScale"])                   # KAMMA.py:238
    c.torque_clip = 4.0                                           # KAMMA.py:649
    c.w1, c.w2, c.w3 = -25.0, -0.02, -0.09                        # KAMMA.py:785
    c.sparse_bonus, c.sparse_threshold = 0.2, 0.01                # KAMMA.py:789-794
    c.terminal_reward = float(learn["terminalReward"])            # KAMMA.py:242
    rng = env["randomCommandVelocityRanges"]
    c.cmd_x_lo, c.cmd_x_hi = map(float, rng["linear_x"])          # KAMMA.py:245
    c.cmd_y_lo, c.cmd_y_hi = map(float, rng["linear_y"])          # KAMMA.py:246
    c.cmd_yaw_lo, c.cmd_yaw_hi = map(float, rng["yaw"])           # KAMMA.py:247
    # torch_rand_float: (upper - lower) is a Python double, rounded to fp32 when it meets the tensor
    c.cmd_x_span = float(rng["linear_x"][1]) - float(rng["linear_x"][0])
    c.cmd_y_span = float(rng["linear_y"][1]) - float(rng["linear_y"][0])
    c.cmd_yaw_span = float(rng["yaw"][1]) - float(rng["yaw"][0])
    init = env["baseInitState"]
    c.init_pos[:] = [float(x) for x in init["pos"]]               # KAMMA.py:251
    c.init_quat[:] = [float(x) for x in init["rot"]]
    c.init_lin_vel[:] = [float(x) for x in init["vLinear"]]
    c.init_ang_vel[:] = [float(x) for x in init["vAngular"]]
    c.init_dof_vel[:] = [float(x) for x in env["dofInitVelocities"]]  # KAMMA.py:274
    c.spawn_xy_range = 0.5                                        # KAMMA.py:424
    c.bounds = 50.0                                               # KAMMA.py:187
    c.clip_obs = float(env.get("clipObservations", math.inf))     # rl_task.py:99
    c.eval_ramp_steps = 50000.0                                   # KAMMA_eval.py:180, 547

    # physics (docs/SPEC.md §3)
    m, com, inertia = composite_body()
    c.gravity = -float(sim["gravity"][2])                         # KAMMA.py:84
    c.body_mass = m
    c.body_inertia[:] = list(inertia)
    c.com_height = com
    c.wheel_pos[:] = list(ROBOT["wheel_pos"])
    c.wheel_radius = float(ctrl.get("wheel_radius", ROBOT["wheel_radius"]))  # KAMMA.py:79
    c.wheel_spin_inertia = ROBOT["wheel_inertia"]
    c.wheel_damping = ROBOT["wheel_damping"]
    c.wheel_limit_velocity = ROBOT["wheel_limit_velocity"]
    c.wheel_limit_damping = ROBOT["wheel_limit_damping"]
    # wheel-ground friction: the knob the scripts turn is sim.default_physics_material.static_friction
    # (train_*.py:66-67, 88-89); terrain.staticFriction (KAMMA.py:51) belongs to the commented-out
    # terrain mesh. The ground plane's own material and PhysX's combine mode are third-party and
    # not visible here (docs/SPEC.md section 3: physics parity unpinned).
    mat = sim.get("default_physics_material", {})
    c.friction = float(mat["static_friction"]) if "static_friction" in mat else float(terrain.get("staticFriction", 0.85))
    c.baumgarte = ROBOT["baumgarte"]
    physx = sim.get("physx", {})
    c.max_depenetration_velocity = float(physx.get("max_depenetration_velocity", 100.0))  # KAMMA.py:105
    c.max_angular_velocity = ROBOT["max_angular_velocity"]
    c.solver_iterations = int(physx.get("solver_position_iteration_count", 4))  # KAMMA.py:96
    for k, v in overrides.items():
        if not hasattr(c, k):
            raise TypeError(f"unknown ois_config field {k!r}")
        if isinstance(v, (list, tuple, np.ndarray)):
            getattr(c, k)[:] = [float(x) for x in v]
        else:
            setattr(c, k, v)
    return c


def config_to_dict(c: _abi.OisConfig) -> Dict[str, Any]:
    out = {}
    for name, _ in c._fields_:
        v = getattr(c, name)
        out[name] = list(v) if hasattr(v, "__len__") else v
    return out
