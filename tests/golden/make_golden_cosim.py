#!/usr/bin/env python3
"""Golden vectors made by running the reference's own task statements in lock-step with the CPU oracle (build
container only; needs /root/reference). Writes

  tests/golden/refexec_cosim.npz        per variant: T steps of n arenas — inputs (actions, the uniform draws, the
                                        simulator's outcome per step) and what the REFERENCE's statements computed
                                        from them (observations, rewards, masks, torques, sums, commands, counters,
                                        extras, the 26 env-0 diagnostics)
  tests/golden/refexec_blocks.npz       single methods on seeded random states (wider numeric ranges than a trajectory
                                        reaches): pre_physics_step of the 12 env files, is_done, reset_idx's extras,
                                        the projection tail, the box-sampler bounds
  tests/golden/refexec_cosim_lines.json which source lines of each method ran and which were removed (refexec_task.py)

How: refexec_task.build() compiles `TorqueDistributionTask` of a reference env file without the statements that need
Isaac Sim (rule R1-R4 there; nothing is written in their place). What those statements would have delivered is DATA
here: the post-physics state comes from the oracle's physics (orc_test_hooks.physics_out), the uniform draws of
`torch_rand_float` from a numpy generator (the same numbers go to the oracle through orc_test_hooks.reset_u),
`torch.rand` / `torch.normal` are the reference's own calls on torch's generator, re-seeded before each call so that
the draw can be recorded and handed to the oracle (switch_u, the GridSampling normal), and the three rotated vectors
of `quat_apply` come from the oracle's orc_quat_apply. tests/test_refexec_cosim.py replays the fixture through the
oracle with physics_in = the recorded simulator outcome, so the fixture pins everything of the step EXCEPT the physics
and does not move when the physics model is re-calibrated.

    python tests/golden/make_golden_cosim.py
"""
import ast
import json
import os
import sys
import types
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import refexec_task as rt  # noqa: E402
from make_golden_ref import SUM_KEYS, _apply_overrides  # noqa: E402

F = torch.float32
EVAL_KEYS = ["Smoothness", "Tracking error", "Desired velocity"]


def task_cfg_literal(variant):
    tree = ast.parse(open(os.path.join(rt.REF, "my_envs", variant + ".py")).read())
    node = next(n for n in tree.body if isinstance(n, ast.Assign) and getattr(n.targets[0], "id", None) == "TASK_CFG")
    run = _apply_overrides(ast.literal_eval(node.value))
    return run


def capture_locals(fn, *args):
    """fn(*args) unchanged; returns (result, the locals of fn's own frame when it returned)"""
    box = {}

    def prof(frame, event, arg):
        if event == "return" and frame.f_code is fn.__code__:
            box.update(frame.f_locals)
    sys.setprofile(prof)
    try:
        res = fn(*args)
    finally:
        sys.setprofile(None)
    return res, box


def rf(lo, hi, u):
    """what the removed torch_rand_float(lo, hi, shape) statements deliver for the uniform draw u (published formula
    of omni.isaac.core.utils.torch.maths: (upper - lower) * torch.rand(shape) + lower)"""
    return (hi - lo) * u + lo


# env-file runs with a TASK_CFG leaf changed (name -> (env file, {path: value})): the terminal reward is 0.0 in every file,
# which hides `terminalReward * reset_buf * ~timeout_buf` (KAMMA.py:801, `~` on an int64 tensor: -1 / -2)
PATCHED = {"KAMMA_medium+terminal1.5": ("KAMMA_medium", {"task/env/learn/terminalReward": 1.5}),
           # config plumbing through the whole step: controller gain, action scale, policy dt = 5 x 0.01 s and an 80-step
           # episode (update_config :258-263), a tight observation clip (rl_task.py:99), and a command range whose small
           # draws are zeroed by the mask of :458
           "KAMMA+config": ("KAMMA", {"task/env/control/stiffness": 0.05, "task/env/control/actionScale": 2.0,
                                      "task/env/control/decimation": 5, "task/env/learn/episodeLength_s": 4.0,
                                      "task/env/clipObservations": 5.0,
                                      "task/env/randomCommandVelocityRanges/linear_x": [0.0, 0.6]}),
           # the per-terrain-level command curriculum (terrain.curriculum: reset_idx skips its re-draw :490, the tail of
           # post_physics_step re-draws the command of EVERY arena after get_observations :719-725, `fraction` :498 is re-snapshot for
           # all arenas by any reset). Arenas on levels 1, 3, 4 and — while their `fraction` keeps torch.normal's sigma positive,
           # i.e. as long as only synchronous timeouts reset them — 0 and 2
           "KAMMA+curriculum": ("KAMMA", {"task/env/terrain/curriculum": True}),
           # the same with robots falling alone: ANY reset re-snapshots `fraction` of every arena (:498) from running sums that are
           # negative, which the generators of levels 0 and 2 cannot take (torch.normal raises on sigma < 0): levels 1, 3, 4 only
           "KAMMA+curriculum_falls": ("KAMMA", {"task/env/terrain/curriculum": True})}
# terrain_levels of the runs' arenas (the reference's are zeros without a terrain)
CURRICULUM_LEVELS = {"KAMMA+curriculum": [1, 3, 4, 1, 3, 0, 2, 1], "KAMMA+curriculum_falls": [1, 3, 4, 4, 1, 3]}


def _patch(cfg, patches):
    for path, value in patches.items():
        d = cfg
        keys = path.split("/")
        for k in keys[:-1]:
            d = d[k]
        assert keys[-1] in d
        d[keys[-1]] = value


class Cosim:
    def __init__(self, variant, n, seed):
        run_name = variant
        variant, patches = PATCHED.get(variant, (variant, {}))
        from options_in_simulation_amd import _abi, build_config
        from oracle.binding import OracleEnv, PHYSICS_STRIDE, RESET_U_STRIDE, fptr, lib
        self._abi, self.L, self.fptr = _abi, lib(), fptr
        self.variant, self.n, self.run_name = variant, n, run_name
        Ref, self.ns, self.lines = rt.build(variant)
        run = task_cfg_literal(variant)
        run["task"]["env"]["numEnvs"] = n
        _patch(run, patches)
        self.cfg = build_config(run, variant=variant, num_envs=n, seed=seed)   # from the reference's own TASK_CFG literal
        from options_in_simulation_amd import default_task_cfg
        mine = default_task_cfg()
        _patch(mine, patches)
        assert bytes(self.cfg) == bytes(build_config(mine, variant=variant, num_envs=n, seed=seed)), "default_task_cfg() differs from the literal"
        self.orc = OracleEnv(self.cfg)
        self.rng = np.random.default_rng(seed)
        self.S = np.zeros((n, PHYSICS_STRIDE), np.float32)
        self.reset_u = np.zeros((n, RESET_U_STRIDE), np.float32)
        self.switch_u = np.zeros(n, np.float32)
        self.grid_seed = None
        self.level_z = np.zeros(n, np.float32)
        self.levels = np.array((CURRICULUM_LEVELS.get(run_name, [0]) * n)[:n], np.int32)
        self.t_index = -1
        co = self

        class Glue(Ref):
            """the caller's side of the removed statements (see the module docstring); no arithmetic of the step here"""

            def reset_idx(self, env_ids):
                ids = env_ids.long()
                u = torch.from_numpy(co.reset_u[ids.numpy()][:, :8].copy())
                # KAMMA.py:448-456 (removed: torch_rand_float) — the draws land where the kept line :458 reads them
                self.commands[ids, 0] = rf(self.command_x_range[0], self.command_x_range[1], u[:, 3])
                self.commands[ids, 1] = rf(self.command_y_range[0], self.command_y_range[1], u[:, 4])
                self.commands[ids, 3] = rf(self.command_yaw_range[0], self.command_yaw_range[1], u[:, 5])
                self._u6 = {int(i): u[k, 6] for k, i in enumerate(ids)}
                super().reset_idx(env_ids)
                # :424, :426-434 (removed: torch_rand_float, quat_mul): spawn offset and yaw quaternion, from the oracle
                for k, i in enumerate(ids.tolist()):
                    pos, quat, cmd = np.zeros(3, np.float32), np.zeros(4, np.float32), np.zeros(4, np.float32)
                    uu = np.ascontiguousarray(co.reset_u[i, :8])
                    co.L.orc_reset_draws(co.cfg, fptr(uu), fptr(pos), fptr(quat), fptr(cmd))
                    assert float(self.base_pos[i, 2]) == float(pos[2]) and float(self.base_pos[i, 0]) == 0.0  # :422 ran
                    self.base_pos[i, 0:2] = torch.from_numpy(pos[:2].copy())
                    self.base_quat[i] = torch.from_numpy(quat.copy())
                co.orientation_givens(self)

            def refresh_dof_state_tensors(self):     # KAMMA.py:587-589 (removed): the attributes are re-bound, as there
                S = torch.from_numpy(co.S.copy())
                self.dof_vel = S[:, 13:17].clone()

            def refresh_body_state_tensors(self):    # :591-593 (removed)
                S = torch.from_numpy(co.S.copy())
                self.base_pos, self.base_quat = S[:, 17:20].clone(), S[:, 3:7].clone()
                self.base_velocities = torch.cat([S[:, 7:10], S[:, 10:13]], 1)
                co.orientation_givens(self)

            def sample_velocity_command(self, env_id):
                i = int(env_id)
                if getattr(self, "evalsampling", False):
                    return super().sample_velocity_command(env_id)
                if self.curriculum:         # KAMMA.py:533-567: torch.normal is the reference's own, on the seed its z was recorded from
                    torch.manual_seed(co.level_seed(i))
                    return super().sample_velocity_command(env_id)
                if self.randsampling:       # :571 removed; :573-575 run on the draw
                    co.ns["x_vel"] = rf(self.command_x_range[0], self.command_x_range[1], self._u6[i]).clone()
                    return super().sample_velocity_command(env_id)
                if self.boxsampling:        # KAMMA_curr1.py:579-591 run; :593 removed; :595-597 run on the draw
                    co.ns["x_vel"] = torch.tensor(0.0)
                    _, loc = capture_locals(super().sample_velocity_command.__func__, self, env_id)
                    co.box_bounds = (loc["x_min"], loc["x_max"])
                    co.ns["x_vel"] = rf(loc["x_min"], loc["x_max"], self._u6[i]).squeeze()
                    return super().sample_velocity_command(env_id)
                if self.gridsampling:       # KAMMA_curr2.py:602-640: torch.rand(1) / torch.normal are the reference's own
                    torch.manual_seed(co.grid_seed)
                    with warnings.catch_warnings():
                        warnings.simplefilter("ignore")
                        return super().sample_velocity_command(env_id)
                raise AssertionError("no sampler")

        t = Glue.__new__(Glue)
        t._device = "cpu"                                                       # RLTask.__init__ (removed): rl_task.py:73
        Glue.__init__(t, "TorqueDistributionTask", types.SimpleNamespace(config=run, task_config=run["task"]), None)
        t.clip_obs = run["task"]["env"].get("clipObservations", np.inf)         # rl_task.py:96
        t.cleanup()                                                             # rl_task.py:118
        t.num_dof = 4                                                           # KAMMA.py:399 (removed: self._robots.num_dof)
        if t.curriculum:
            t.terrain_levels = torch.from_numpy(self.levels.astype(np.int64))   # KAMMA.py:223 has zeros: no terrain to take levels from
        self.task = t
        self.gravity = np.array([0.0, 0.0, -1.0], np.float32)

    def level_seed(self, i):
        return 9000 + (self.t_index + 1) * self.n + i

    def draw_level_z(self):
        """the standard normal behind each arena's torch.normal of this step's re-draw, by replaying its seed"""
        for i in range(self.n):
            torch.manual_seed(self.level_seed(i))
            self.level_z[i] = torch.randn(1)[0]

    # -- data the removed statements would have delivered ---------------------------------------------------------
    def quat_apply_all(self, quat, vec):
        out = np.zeros((self.n, 3), np.float32)
        q = np.ascontiguousarray(quat, np.float32)
        v = np.ascontiguousarray(vec, np.float32)
        for i in range(self.n):
            self.L.orc_quat_apply(self.fptr(q[i]), self.fptr(v), self.fptr(out[i]))
        return torch.from_numpy(out)

    def orientation_givens(self, t):
        q = t.base_quat.numpy()
        self.ns["projected_gravity"] = self.quat_apply_all(q, self.gravity)                    # KAMMA.py:749
        self.ns["forward"] = self.quat_apply_all(q, np.array([1.0, 0.0, 0.0], np.float32))      # :701
        self.ns["up_dir"] = self.quat_apply_all(q, np.array([0.0, 0.0, 1.0], np.float32))       # :707

    def draw_reset(self, t_index):
        self.reset_u[:] = self.rng.random(self.reset_u.shape, np.float32)
        if self.task.gridsampling and not self.task.randsampling and not self.task.boxsampling:
            assert self.n == 1
            self.grid_seed = 5000 + t_index
            torch.manual_seed(self.grid_seed)
            self.reset_u[0, 6] = torch.rand(1)[0]
            self.reset_u[0, 8] = torch.randn(1)[0]

    # -- the run --------------------------------------------------------------------------------------------------
    def start(self, seed):
        self.draw_reset(-1)
        self.first_reset_u = self.reset_u.copy()
        self.orc.set_hooks(reset_u=self.reset_u, **(dict(levels=self.levels) if self.task.curriculum else {}))
        self.orc.reset(seed)
        self.task.post_reset()
        return self.snapshot()

    def step(self, t_index, actions, events=()):
        A = self._abi
        T = self.task
        for kind, arena in events:
            if kind == "fall":
                self.orc.buffer(A.BUF_QUAT)[arena] = np.array([0.0, 1.0, 0.0, 0.0], np.float32)
            elif kind == "oob":
                self.orc.buffer(A.BUF_POSE)[arena, 0] += 60.0
            elif kind == "oob_y":
                self.orc.buffer(A.BUF_POSE)[arena, 1] -= 70.0
        torch.manual_seed(1000 + t_index)
        self.switch_u[:] = torch.rand(self.n, 1)[:, 0].numpy()   # the draw of KAMMA.py:633, recorded by replaying the seed
        torch.manual_seed(1000 + t_index)
        T.pre_physics_step(torch.from_numpy(actions.copy()))
        pre = {"v_delta": T.v_delta.clone(), "omega_delta": T.omega_delta.clone(), "ac_left": T.ac_left.clone(),
               "ac_right": T.ac_right.clone(), "torques": T.torques.clone(), "wheel_torqs": T.wheel_torqs.clone(),
               "guiding": T.guiding_reward.clone()}
        self.draw_reset(t_index)
        self.t_index = t_index
        lv = {}
        if T.curriculum:
            self.draw_level_z()
            lv = dict(level_z=self.level_z, levels=self.levels)
        self.orc.set_hooks(switch_u=self.switch_u, reset_u=self.reset_u, physics_out=self.S, **lv)
        self.orc.step(actions)
        T.post_physics_step()   # its refresh_*_state_tensors calls bind the simulator's outcome (Glue)
        out = self.snapshot()
        out.update({k: v.numpy() for k, v in pre.items()})
        used = self.reset_u.copy()
        used[~out["reset"].astype(bool)] = 0.0   # only a resetting arena reads its draws: the others' are not stored
        out.update(actions=actions.copy(), switch_u=self.switch_u.copy(), reset_u=used, physics=self.S[:, :17].copy())
        if T.curriculum:
            out.update(level_z=self.level_z.copy(), levels=self.levels.copy())
        # the projection's 3-term sum is order dependent (1 ulp); both sides continue from the oracle's value
        orc_vfwd = torch.from_numpy(self.orc.buffer(A.BUF_POSE)[:, 3].copy())
        assert torch.allclose(T.v_forward_projected, orc_vfwd, rtol=2e-6, atol=2e-7), (T.v_forward_projected, orc_vfwd)
        T.v_forward_projected = orc_vfwd
        return out

    def snapshot(self):
        T = self.task
        sums = torch.stack([T.episode_sums[k] for k in SUM_KEYS], 1)
        d = {"obs": T.obs_buf.clone(), "rew": T.rew_buf.clone(), "reset": T.reset_buf.to(torch.uint8), "timeout": T.timeout_buf.to(torch.uint8),
             "sums": sums, "commands": T.commands.clone(), "episode_buf": T.episode_buf.to(torch.int32), "sim_steps": T.sim_steps.to(torch.int32),
             "episode_count": T.episode_count.to(torch.int32), "prev_lin": T.previous_linear_velocity[:, 0].clone(),
             "prev_ang": T.previous_angular_velocity[:, 2].clone(), "v_fwd": T.v_forward_projected.clone(),
             "v_lat": T.v_lateral_projected.clone(), "v_up": T.v_upward_projected.clone()}
        if all(k in T.episode_sums for k in EVAL_KEYS):
            d["sums_eval"] = torch.stack([T.episode_sums[k] for k in EVAL_KEYS], 1)
            d["old_torques"] = T.old_torques.clone()
        for g in ("gamma_assist1", "gamma_assist2", "gamma_assist"):
            if hasattr(T, g):
                d[g] = T.__dict__[g].clone().float()
        ex = T.extras.get("episode", {})
        d["extras_keys"] = np.array(sorted(ex))
        d["extras_order"] = np.array(list(ex))  # the ORDER the reference's reset_idx fills the dict in (what a logger iterates)
        d["extras"] = np.array([float(ex[k]) for k in sorted(ex)], np.float64)
        oc = getattr(T, "observed_components", None)
        if oc:
            d["components_keys"] = np.array(list(oc))
            d["components"] = np.array([float(oc[k]) for k in oc], np.float64)
        if hasattr(T, "fraction"):
            d["fraction"] = T.fraction.clone()
        return {k: (v.numpy() if isinstance(v, torch.Tensor) else v) for k, v in d.items()}


RUNS = {  # variant -> (arenas, steps, {step: [(event, arena)]}); two synchronous timeouts need 199 steps
    "KAMMA": (6, 204, {20: [("fall", 3)], 21: [("fall", 3)], 60: [("fall", 5)], 75: [("oob", 1)], 130: [("oob_y", 0), ("fall", 2)]}),
    "KAMMA_medium": (4, 104, {33: [("fall", 1)]}),
    "KAMMA_g11_g21": (4, 104, {33: [("oob", 2)]}),
    "KAMMA_g1d_g21": (4, 104, {40: [("fall", 0)]}),
    "KAMMA_curr1": (1, 204, {50: [("fall", 0)]}),     # BoxSampling / GridSampling only run with one arena (min(1.0, tensor))
    "KAMMA_curr2": (1, 204, {50: [("oob", 0)]}),
    "KAMMA_eval": (5, 130, {20: [("fall", 3)], 21: [("fall", 3)], 60: [("fall", 4)]}),
    "KA_DDPG_4D": (4, 130, {20: [("fall", 3)], 45: [("oob", 0)]}),
    "KA_DDPG_2D": (4, 104, {20: [("fall", 3)]}),
    "KA_DDPG_1D": (4, 104, {20: [("fall", 1)]}),
    "KA_DDPG_2D_eval": (3, 104, {20: [("fall", 1)]}),
    "KA_DDPG_1D_eval": (3, 104, {30: [("fall", 2)]}),
    "KAMMA_medium+terminal1.5": (4, 104, {33: [("fall", 1)], 50: [("oob", 2)]}),
    "KAMMA+config": (6, 170, {33: [("fall", 1)], 50: [("oob", 2)]}),
    "KAMMA+curriculum": (8, 120, {}),
    "KAMMA+curriculum_falls": (6, 140, {20: [("fall", 2)], 21: [("fall", 2)], 60: [("oob", 4)]}),
}
# sim_steps where the gamma schedule is in the interior (the trajectories above start at 0: gamma 1 - 2e-4)
START_SIM_STEPS = {"KAMMA+config": 250000, "KAMMA": 150000, "KAMMA_medium": 420000, "KAMMA_medium+terminal1.5": 100000, "KAMMA_g1d_g21": 250000, "KAMMA_curr1": 200000, "KAMMA_curr2": 300000,
                   "KA_DDPG_4D": 350000, "KAMMA_eval": 20000, "KA_DDPG_2D_eval": 30000, "KA_DDPG_1D_eval": 49950}


def run_variant(variant, n, T, seed, events):
    from tests.cosim_check import check_step
    from options_in_simulation_amd import _abi
    co = Cosim(variant, n, seed)
    first = co.start(seed)
    drawn = bool(co.task.curriculum)
    check_step(co.orc, co.cfg, first, f"{variant} post_reset", first=True, drawn_commands=drawn)
    if variant in START_SIM_STEPS:   # a run that has been training for a while: same value on both sides
        co.task.sim_steps[:] = START_SIM_STEPS[variant]
        co.orc.buffer(_abi.BUF_COUNTERS)[:, 1] = START_SIM_STEPS[variant]
    rng = np.random.default_rng(seed + 1)
    lo = -1.2 if variant.startswith("KA_DDPG") else -0.2
    steps = []
    for t in range(T):
        a = rng.uniform(lo, 1.5, (n, co.cfg.action_dim)).astype(np.float32)
        try:
            out = co.step(t, a, events.get(t, ()))
        except ZeroDivisionError:
            # KAMMA_g11_g21.py:859: 100.0 * (1 - gamma2) * comp / rew_buf[0].item() in Python floats. With gamma2 = 1 and the
            # expert's torques inside the clip the reward is exactly 0 and the reference stops here (its sibling
            # KAMMA_g1d_g21 has the guard :838-842 for this). Recorded: the step, and that the oracle's reward is 0 there.
            assert co.orc.rew[0] == 0.0 and variant == "KAMMA_g11_g21"
            co.raised_at = t
            break
        check_step(co.orc, co.cfg, out, f"{variant} step {t}", drawn_commands=drawn)
        steps.append(out)
    return co, first, steps


def pack(variant, co, first, steps):
    d = {}
    per_step = [k for k in steps[0] if k not in ("extras_keys", "extras_order", "components_keys")]
    for k in per_step:
        d[k] = np.stack([s[k] for s in steps])
    d["extras_keys"], d["components_keys"] = steps[0]["extras_keys"], steps[0]["components_keys"]
    d["extras_order"] = steps[0]["extras_order"]
    assert all((s["extras_keys"] == d["extras_keys"]).all() and (s["components_keys"] == d["components_keys"]).all()
               and (s["extras_order"] == d["extras_order"]).all() for s in steps)
    for k in ("commands", "episode_buf", "sim_steps", "episode_count", "extras"):
        d["first__" + k] = first[k]
    d["first__reset_u"] = co.first_reset_u
    d["start_sim_steps"] = np.int32(START_SIM_STEPS.get(variant, 0))
    d["reference_raised_zero_division_at"] = np.int32(getattr(co, "raised_at", -1))
    d["env_file"] = np.array(co.variant)
    d["task_cfg_patches_json"] = np.array(json.dumps(PATCHED.get(variant, (variant, {}))[1], sort_keys=True))
    return {f"{variant}__{k}": v for k, v in d.items()}


# ------------------------------------------------------------------------------------------------------------------
# single methods on seeded random states (wider ranges than a trajectory reaches)
def _rand(g):
    return lambda *s, lo=-1.0, hi=1.0: (hi - lo) * torch.rand(*s, generator=g, dtype=F) + lo


def block_prephysics(variant, n=129, seed=0):
    """pre_physics_step of the file on random commands / speeds / actions, sim_steps from 0 to 1.3 max_sim_steps (the
    clamp of the schedule at 0), actions beyond the action space (the clip). The torch.rand of the switch is the
    reference's own; its draw is recorded by replaying the seed."""
    co = Cosim(variant, n, 300 + seed)   # __init__ / update_config only: post_reset's samplers of curr1 / curr2 need one arena
    T = co.task
    g = torch.Generator().manual_seed(seed)
    r = _rand(g)
    T.commands = torch.stack([r(n, lo=0.0, hi=2.0), r(n, lo=-0.5, hi=0.5), r(n, lo=-0.3, hi=0.3), r(n, lo=1.0, hi=1.1)], 1)
    T.v_forward_projected = r(n, lo=-0.5, hi=2.5)
    T.base_velocities = r(n, 6)
    T.sim_steps = torch.randint(0, int(1.3 * T.max_sim_steps), (n,), generator=g)
    if hasattr(T, "old_torques"):
        T.old_torques = torch.clip(r(n, 4, lo=-5.0, hi=5.0), -4.0, 4.0)
        for k in ("Smoothness", "Desired velocity"):
            T.episode_sums[k] = r(n, lo=0.0, hi=3.0)
    A = co.cfg.action_dim
    actions = r(n, A, lo=-1.3, hi=1.7)
    inp = {"commands": T.commands.clone(), "v_forward_projected": T.v_forward_projected.clone(), "omega_z": T.base_velocities[:, 5].clone(),
           "sim_steps": T.sim_steps.to(torch.int32), "actions": actions.clone()}
    if hasattr(T, "old_torques"):
        inp["old_torques"] = T.old_torques.clone()
        inp["sums_eval_in"] = torch.stack([T.episode_sums["Smoothness"], T.episode_sums["Desired velocity"]], 1).clone()
    torch.manual_seed(77 + seed)
    inp["switch_u"] = torch.rand(n, 1)[:, 0]
    torch.manual_seed(77 + seed)
    T.pre_physics_step(actions)
    out = {"v_delta": T.v_delta, "omega_delta": T.omega_delta, "ac_left": T.ac_left, "ac_right": T.ac_right, "torques": T.torques,
           "wheel_torqs": T.wheel_torqs, "guiding": T.guiding_reward}
    for gk in ("gamma_assist1", "gamma_assist2", "gamma_assist"):
        if hasattr(T, gk):
            out[gk] = T.__dict__[gk].float()
    if hasattr(T, "old_torques"):
        out["old_torques_out"] = T.old_torques
        out["sums_eval_out"] = torch.stack([T.episode_sums["Smoothness"], T.episode_sums["Desired velocity"]], 1)
    # the build's own transcription (make_golden.prephysics_formulas, what refmath_step.npz was made with) on the same inputs
    from tests.golden.make_golden import prephysics_formulas
    a4 = actions if A == 4 else (actions.repeat_interleave(2, dim=1) if A == 2 else actions.repeat(1, 4))
    tr = prephysics_formulas(inp["commands"], inp["v_forward_projected"], inp["omega_z"], a4, T.sim_steps, inp["switch_u"], Kp=T.Kp, Kp_omega=T.Kp_omega,
                             vehicle_mass=T.vehicle_mass, vehicle_inertia=T.vehicle_inertia, dt=T.dt, action_scale=T.action_scale,
                             max_sim_steps=T.max_sim_steps)
    kind = "blend" if variant.startswith("KA_DDPG") else "switch"
    for mine, theirs in (("v_delta", "v_delta"), ("omega_delta", "omega_delta"), ("ac_left", "ac_left"), ("ac_right", "ac_right")):
        assert torch.equal(tr[mine], out[theirs]), (variant, mine)
    const_gamma = co.cfg.gamma1_mode == co._abi.GAMMA_CONST
    if not const_gamma:   # with the schedule lines commented out the files keep their constructor value: nothing to transcribe
        assert torch.equal(tr["gamma"], out.get("gamma_assist1", out.get("gamma_assist"))), variant
        assert torch.equal(tr[kind + "_torques"], out["torques"]) and torch.equal(tr[kind + "_wheel_torqs"], out["wheel_torqs"]), variant
        guide = "blend_guiding" if kind == "blend" else "switch_guiding"   # (KAMMA_curr1 / _curr2 set the policy-torque form
        # first and overwrite it after the loop with the executed-torque one, like KAMMA.py:656)
        assert torch.equal(tr[guide], out["guiding"]), (variant, guide)
    d = {k: v.numpy() for k, v in {**inp, **out}.items()}
    d["max_sim_steps"] = np.float32(T.max_sim_steps)
    return d


def block_is_done(n=257, seed=1):
    """is_done (KAMMA.py:736-764) with projected_gravity handed in (the quat_apply of :749 is removed)"""
    co = Cosim("KAMMA", n, 400)
    co.start(400)
    T = co.task
    g = torch.Generator().manual_seed(seed)
    r = _rand(g)
    q = torch.nn.functional.normalize(r(n, 4), dim=1)
    q[: n // 3] = torch.nn.functional.normalize(torch.tensor([[1.0, 0.02, -0.03, 0.4]]) + 0.05 * r(n // 3, 4), dim=1)
    T.base_quat = q
    T.base_pos = r(n, 3, lo=-60.0, hi=60.0)
    T.base_pos[::2] *= 0.1
    T.base_pos[5, 0], T.base_pos[6, 0], T.base_pos[7, 1], T.base_pos[8, 1] = 50.0, -50.0, 50.0, -50.0   # on the bounds: inside
    T.episode_buf = torch.randint(0, 120, (n,), generator=g)
    T.episode_buf[:4] = torch.tensor([98, 99, 100, 97])
    co.orientation_givens(T)
    T.is_done()
    return {"quat": q.numpy(), "base_pos": T.base_pos.numpy(), "episode_buf": T.episode_buf.to(torch.int32).numpy(),
            "projected_gravity_z": co.ns["projected_gravity"][:, 2].numpy(), "reset": T.reset_buf.to(torch.uint8).numpy(),
            "timeout": T.timeout_buf.to(torch.uint8).numpy(), "has_fallen": T.has_fallen.to(torch.uint8).numpy(),
            "out_of_bounds": T.out_of_bounds.to(torch.uint8).numpy(), "max_episode_length": np.int32(T.max_episode_length)}


def block_reset_extras(variant, n=64, seed=2, gamma=0.375):
    """reset_idx (:458-488, :498) on random episode sums — some with a zero or tiny final sum (inf / nan ratios) — for a
    subset of the arenas: counters, the ratio sums under their NEW keys (:470-476), the means of extras, `fraction`."""
    co = Cosim(variant, n, 500 + seed)
    co.start(500 + seed)
    T = co.task
    g = torch.Generator().manual_seed(seed)
    sums = 10.0 * (torch.rand(n, 8, generator=g, dtype=F) - 0.7)
    sums[3, 7] = 0.0; sums[9, 7] = 1e-30; sums[12] = 0.0
    for j, k in enumerate(SUM_KEYS):
        T.episode_sums[k] = sums[:, j].clone()
    single = hasattr(T, "gamma_assist") and not hasattr(T, "gamma_assist2")
    if single:
        T.gamma_assist = torch.full((n,), gamma, dtype=F)
    else:
        T.gamma_assist1, T.gamma_assist2 = torch.full((n,), 0.25, dtype=F), torch.full((n,), gamma, dtype=F)
    T.episode_buf = torch.randint(0, 100, (n,), generator=g)
    T.episode_count = torch.randint(0, 50, (n,), generator=g)
    ids = torch.arange(0, n, 3)
    ids = ids[ids != 12]
    before = {"episode_buf": T.episode_buf.clone(), "episode_count": T.episode_count.clone()}
    co.draw_reset(0)
    T.reset_idx(ids)
    ex = T.extras["episode"]
    keys = sorted(ex)
    ratio_keys = ["r1/Final reward", "r2/Final reward", "r3/Final reward", "Dense/Final reward", "Sparse/Final reward", "Guiding/Final reward",
                  "Observed/Final reward"]
    # the ratio tensors are computed for EVERY arena before the sums of `ids` are zeroed; only ids' entries are then zeroed (:485)
    return {"sums": sums.numpy(), "gamma2": np.float32(gamma), "gamma1": np.float32(gamma if single else 0.25), "ids": ids.numpy().astype(np.int32),
            "episode_buf_in": before["episode_buf"].to(torch.int32).numpy(), "episode_count_in": before["episode_count"].to(torch.int32).numpy(),
            "episode_buf_out": T.episode_buf.to(torch.int32).numpy(), "episode_count_out": T.episode_count.to(torch.int32).numpy(),
            "extras_keys": np.array(keys), "extras": np.array([float(ex[k]) for k in keys], np.float64),
            "sums_after": torch.stack([T.episode_sums[k] for k in SUM_KEYS], 1).numpy(),
            "ratios_after": torch.stack([T.episode_sums[k] for k in ratio_keys], 1).numpy(),
            "fraction": T.fraction.numpy(), "threshold_high": np.float32(T.threshold_high), "reset_u": co.reset_u.copy(),
            "commands": T.commands.numpy(), "episode_length_s": np.float32(T.max_episode_length_s)}


def block_box_bounds():
    """KAMMA_curr1.py:579-591 over the whole schedule (one arena: `min(1.0, tensor)`), x_min / x_max read from the frame"""
    co = Cosim("KAMMA_curr1", 1, 600)
    co.start(600)
    T = co.task
    steps = np.concatenate([np.linspace(0, 620000, 60).astype(np.int64), [1, 499999, 500000, 500001]])
    lo, hi, cmd = [], [], []
    for k, st in enumerate(steps):
        T.sim_steps = torch.tensor([int(st)])
        co.draw_reset(k)
        T._u6 = {0: torch.from_numpy(co.reset_u[:, 6].copy())[0]}
        x, om = T.sample_velocity_command(0)
        lo.append(float(co.box_bounds[0])); hi.append(float(co.box_bounds[1])); cmd.append([float(x), float(om), float(co.reset_u[0, 6])])
    return {"sim_steps": steps.astype(np.int32), "x_min": np.array(lo, np.float32), "x_max": np.array(hi, np.float32),
            "command_u": np.array(cmd, np.float32), "max_sim_steps": np.float32(T.max_sim_steps), "x_range": np.array(T.command_x_range, np.float32)}


def block_get_heights(n=96, seed=4):
    """get_heights (KAMMA.py:872-893) on the field of the reference's OWN terrain generator (origin_terrain_generator.Terrain,
    executed: make_golden_ref.terrain_namespace). `points` — the scan lattice after quat_apply_yaw (:874-879, removed: omni's
    normalize / quat_apply) plus the base position — is handed in from the build's transcription (make_golden.refmath_heights'
    formulas); the border shift, the division by the horizontal scale, the truncation, the index clipping, the two gathers,
    their minimum and the vertical scale are the reference's statements."""
    from make_golden_ref import terrain_namespace
    ns_t = terrain_namespace()
    np.random.seed(3)
    terr = ns_t["Terrain"]({"mapLength": 4.0, "mapWidth": 4.0, "numLevels": 3, "numTerrains": 3, "slopeTreshold": 0.5,
                            "terrainProportions": [0.1, 0.5, 0.4], "curriculum": False}, n)   # randomized_terrain: every family
    co = Cosim("KAMMA", n, 700)
    T = co.task
    T.terrain = terr
    T.height_samples = torch.tensor(terr.heightsamples).view(terr.tot_rows, terr.tot_cols)
    g = torch.Generator().manual_seed(seed)
    base_pos = torch.rand((n, 3), generator=g) * torch.tensor([14.0, 14.0, 1.0]) - torch.tensor([1.0, 1.0, 0.0])   # the 3 x 3 tiles of 4 m
    base_pos[:8] = torch.tensor([[-25.0, 1.0, 0], [40.0, 1.0, 0], [1.0, -29.0, 0], [1.0, 40.0, 0], [-20.0, -20.0, 0], [31.9, 31.9, 0],
                                 [-19.99, 3.0, 0], [3.0, 31.95, 0]])                                          # beyond the field: index clipping
    base_quat = torch.nn.functional.normalize(torch.randn((n, 4), generator=g), dim=1)
    base_quat[::3, 1:3] *= 0.05
    base_quat = torch.nn.functional.normalize(base_quat, dim=1)
    # --- the build's transcription of init_height_points (:278-292, commented out in the reference) and quat_apply_yaw (:895-899)
    y = 0.1 * torch.tensor([-5, -4, -3, -2, -1, 1, 2, 3, 4, 5])
    x = 0.1 * torch.tensor([-8, -7, -6, -5, -4, -3, -2, 2, 3, 4, 5, 6, 7, 8])
    grid_x, grid_y = torch.meshgrid(x, y, indexing="ij")
    npts = grid_x.numel()
    height_points = torch.zeros(n, npts, 3)
    height_points[:, :, 0] = grid_x.flatten()
    height_points[:, :, 1] = grid_y.flatten()
    quat_yaw = base_quat.repeat(1, npts).clone().view(-1, 4)
    quat_yaw[:, 1:3] = 0.0
    quat_yaw = quat_yaw / quat_yaw.norm(p=2, dim=-1).clamp(min=1e-9, max=None).unsqueeze(-1)
    b = height_points.reshape(-1, 3)
    xyz = quat_yaw[:, 1:]
    t = xyz.cross(b, dim=-1) * 2
    rotated = (b + quat_yaw[:, :1] * t + xyz.cross(t, dim=-1)).view(height_points.shape)
    co.ns["points"] = rotated + base_pos.unsqueeze(1)
    T.base_pos, T.base_quat, T.num_height_points, T.height_points = base_pos, base_quat, npts, height_points
    measured = T.get_heights()
    return {"height_samples": T.height_samples.numpy(), "base_pos": base_pos.numpy(), "base_quat": base_quat.numpy(),
            "horizontal_scale": np.float32(terr.horizontal_scale), "vertical_scale": np.float32(terr.vertical_scale),
            "border_size": np.float32(terr.border_size), "measured_heights": measured.numpy()}


def blocks():
    out = {}
    for key, val in block_get_heights().items():
        out[f"heights__{key}"] = val
    for k, v in enumerate(v for v in RUNS if v not in PATCHED):
        for key, val in block_prephysics(v, seed=k).items():
            out[f"prephysics__{v}__{key}"] = val
    for key, val in block_is_done().items():
        out[f"is_done__{key}"] = val
    for v in ("KAMMA", "KA_DDPG_4D", "KAMMA_eval"):
        for key, val in block_reset_extras(v).items():
            out[f"reset_extras__{v}__{key}"] = val
    for key, val in block_box_bounds().items():
        out[f"box__{key}"] = val
    np.savez_compressed(os.path.join(HERE, "refexec_blocks.npz"), **out)
    print("wrote refexec_blocks.npz", os.path.getsize(os.path.join(HERE, "refexec_blocks.npz")), "bytes")


def main():
    blocks()
    fixture, report = {}, {}
    for k, variant in enumerate(RUNS):
        n, T, events = RUNS[variant]
        co, first, steps = run_variant(variant, n, T, 100 + k, events)
        fixture.update(pack(variant, co, first, steps))
        report[variant] = rt.lines_report(co.lines)
        resets = np.stack([s["reset"] for s in steps])
        print(f"{variant}: {len(steps)} steps x {n} arenas, {int(resets.sum())} resets ({int(np.stack([s['timeout'] for s in steps]).sum())} timeouts), "
              f"reference == oracle at every step")
    np.savez_compressed(os.path.join(HERE, "refexec_cosim.npz"), **fixture)
    json.dump(report, open(os.path.join(HERE, "refexec_cosim_lines.json"), "w"), indent=1)
    print("wrote refexec_cosim.npz", os.path.getsize(os.path.join(HERE, "refexec_cosim.npz")), "bytes")


if __name__ == "__main__":
    main()
