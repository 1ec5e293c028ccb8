#!/usr/bin/env python3
"""Generates the golden fixtures under tests/golden/ (run from the repo root):

    python tests/golden/make_golden.py

1. refmath_*.npz — the REFERENCE-FOLLOWED math. The reference's env module cannot be imported
   (it needs Isaac Sim / gym / pxr: ModuleNotFoundError), so the formulas of
   src/my_envs/KAMMA.py (and KA_DDPG_4D.py) are transcribed here line by line into torch-CPU
   float32 tensor expressions — one torch op per reference op, same operand order — and
   evaluated on seeded random inputs. Only inputs and outputs are stored. The transcription is
   build code written for this repo; each block cites the reference lines it follows. The
   quaternion helpers restate omni.isaac.core.utils.torch.rotations (third-party, not in the
   reference tree) from the published isaacgym torch_utils formulas of the same names.
2. traj_*.npz — ORACLE trajectories (CPU oracle, oracle/ois_oracle.c) for BASELINE.json
   configs 1/2 in miniature plus the expert-only run: the HIP kernel must reproduce them
   (tests/test_gpu_golden.py), and the oracle must keep reproducing them (regression pin).
   These pin oracle <-> kernel parity only: the physics itself is parity-unpinned.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

F = torch.float32


# ----------------------------------------------------------------------------------------
# third-party helpers restated (wxyz quaternions)
def quat_apply(a, b):
    xyz = a[:, 1:]
    t = xyz.cross(b, dim=-1) * 2
    return b + a[:, :1] * t + xyz.cross(t, dim=-1)


def quat_mul(a, b):
    w1, x1, y1, z1 = a[:, 0], a[:, 1], a[:, 2], a[:, 3]
    w2, x2, y2, z2 = b[:, 0], b[:, 1], b[:, 2], b[:, 3]
    ww = (z1 + x1) * (x2 + y2)
    yy = (w1 - y1) * (w2 + z2)
    zz = (w1 + y1) * (w2 - z2)
    xx = ww + yy + zz
    qq = 0.5 * (xx + (z1 - x1) * (x2 - y2))
    w = qq - ww + (z1 - y1) * (y2 - z2)
    x = qq - xx + (x1 + w1) * (x2 + w2)
    y = qq - yy + (w1 - x1) * (y2 + z2)
    z = qq - zz + (z1 + y1) * (w2 - x2)
    return torch.stack([w, x, y, z], dim=-1)


def torch_rand_float_from(lower, upper, u):
    """torch_rand_float(lower, upper, shape) = (upper - lower) * torch.rand(shape) + lower, with the
    uniform draw `u` passed in (the fixtures carry the draws so any RNG can replay them)."""
    return (upper - lower) * u + lower


def prephysics_formulas(commands, v_forward_projected, omega_z, actions4, sim_steps, rand_vals, *, Kp, Kp_omega, vehicle_mass,
                        vehicle_inertia, dt, action_scale, max_sim_steps):
    """the transcription of pre_physics_step's arithmetic on given inputs (torch float32 tensors; actions4 already [N, 4]):
    KAMMA.py:604-636, :649, :656 and KA_DDPG_4D.py:627-631, :637. make_golden_cosim.py asserts that it equals what the
    reference's own statements compute on the same inputs."""
    out = {}
    # ---- KAMMA.py:604-624 criteria controller ----
    desired_v = commands[:, 0]
    desired_omega = commands[:, 2]
    current_v = v_forward_projected
    current_omega = omega_z
    v_delta = desired_v - current_v
    omega_delta = desired_omega - current_omega
    ac_left = Kp * (vehicle_mass * (v_delta / dt)) - Kp_omega * (vehicle_inertia * (omega_delta / dt))
    ac_right = Kp * (vehicle_mass * (v_delta / dt)) + Kp_omega * (vehicle_inertia * (omega_delta / dt))
    criteria_action = torch.stack([ac_left, ac_left, ac_right, ac_right], dim=1)
    out.update(v_delta=v_delta, omega_delta=omega_delta, ac_left=ac_left, ac_right=ac_right)

    # ---- KAMMA.py:627-628 gamma schedule ----
    gamma_assist = torch.clamp(1.0 - (sim_steps.float() / max_sim_steps), min=0.0)
    out.update(gamma=gamma_assist)

    # ---- KAMMA.py:632-636 stochastic switch, :649 clip, :656 guiding reward ----
    gamma = gamma_assist.view(-1, 1)
    mask = (rand_vals.view(-1, 1) < gamma).float()
    execution_action = mask * criteria_action + (1 - mask) * (actions4 * action_scale)
    wheel_torqs = torch.clip(execution_action, -4.0, 4.0)
    guiding_reward = -torch.norm(wheel_torqs - criteria_action, dim=1)
    out.update(switch_torques=execution_action, switch_wheel_torqs=wheel_torqs, switch_guiding=guiding_reward)

    # ---- KA_DDPG_4D.py:627-631 deterministic blend, :637 guiding on the unclipped policy torque ----
    blend = (torch.tensor(1.0) - gamma) * actions4 * action_scale + gamma * criteria_action
    blend_wheel = torch.clip(blend, -4.0, 4.0)
    blend_guiding = -torch.norm(actions4 * action_scale - criteria_action, dim=1)
    out.update(blend_torques=blend, blend_wheel_torqs=blend_wheel, blend_guiding=blend_guiding)
    return out


# ----------------------------------------------------------------------------------------
def refmath_step(seed=0, n=257):
    g = torch.Generator().manual_seed(seed)
    r = lambda *s, lo=-1.0, hi=1.0: (hi - lo) * torch.rand(*s, generator=g, dtype=F) + lo  # noqa: E731
    out = {}
    # ---- inputs ----
    commands = torch.stack([r(n, lo=0.0, hi=2.0), r(n, lo=-0.5, hi=0.5), r(n, lo=-0.3, hi=0.3), r(n, lo=1.0, hi=1.1)], 1)
    v_forward_projected = r(n, lo=-0.5, hi=2.5)
    base_velocities = r(n, 6, lo=-1.0, hi=1.0)
    actions = r(n, 4, lo=-0.2, hi=1.6)          # beyond [0,1] so the +-4 clip is exercised
    sim_steps = torch.randint(0, 600000, (n,), generator=g)
    rand_vals = torch.rand(n, 1, generator=g, dtype=F)
    Kp, Kp_omega, vehicle_mass, vehicle_inertia, dt, action_scale = 0.035, 0.665, 25.0, 1.05, 0.1, 3.0
    max_sim_steps = 400000.0
    out.update(commands=commands, v_forward_projected=v_forward_projected, base_velocities=base_velocities,
               actions=actions, sim_steps=sim_steps.to(torch.int32), rand_vals=rand_vals[:, 0])

    pre = prephysics_formulas(commands, v_forward_projected, base_velocities[:, 5], actions, sim_steps, rand_vals[:, 0], Kp=Kp,
                              Kp_omega=Kp_omega, vehicle_mass=vehicle_mass, vehicle_inertia=vehicle_inertia, dt=dt, action_scale=action_scale,
                              max_sim_steps=max_sim_steps)
    out.update(pre)
    v_delta, omega_delta, gamma_assist = pre["v_delta"], pre["omega_delta"], pre["gamma"]
    wheel_torqs, guiding_reward = pre["switch_wheel_torqs"], pre["switch_guiding"]
    desired_v, desired_omega = commands[:, 0], commands[:, 2]

    # ---- KAMMA.py:736-764 is_done ----
    q = torch.nn.functional.normalize(r(n, 4), dim=1)
    q[: n // 3] = torch.nn.functional.normalize(torch.tensor([[1.0, 0.02, -0.03, 0.4]]) + 0.05 * r(n // 3, 4), dim=1)
    base_pos = r(n, 3, lo=-60.0, hi=60.0)
    base_pos[::2] *= 0.1
    episode_buf = torch.randint(0, 120, (n,), generator=g)
    max_episode_length = 100
    timeout_buf = torch.where(episode_buf >= max_episode_length - 1, torch.ones(n, dtype=torch.long),
                              torch.zeros(n, dtype=torch.long))
    gravity_vec = torch.tensor([0.0, 0.0, -1.0]).repeat((n, 1))
    projected_gravity = quat_apply(q, gravity_vec)
    has_fallen = projected_gravity[:, 2] > 0.0
    reset_buf = has_fallen.clone()
    reset_buf = torch.where(timeout_buf.bool(), torch.ones_like(reset_buf), reset_buf)
    bounds = torch.tensor([-50.0, 50.0, -50.0, 50.0])
    out_of_bounds = (base_pos[:, 0] < bounds[0]) | (base_pos[:, 0] > bounds[1]) | \
                    (base_pos[:, 1] < bounds[2]) | (base_pos[:, 1] > bounds[3])
    reset_buf = torch.where(out_of_bounds, torch.ones_like(reset_buf), reset_buf)
    out.update(quat=q, base_pos=base_pos, episode_buf=episode_buf.to(torch.int32),
               timeout=timeout_buf.to(torch.uint8), reset=reset_buf.to(torch.uint8))

    # ---- KAMMA.py:767-822 calculate_metrics (terminalReward 0.0 and 1.5: the ~timeout quirk) ----
    previous_linear_velocity = r(n, lo=-0.5, hi=2.5)
    previous_angular_velocity = r(n)
    gamma2 = gamma_assist
    episode_sums = r(n, 8, lo=-5.0, hi=5.0)
    out.update(prev_lin_in=previous_linear_velocity.clone(), prev_ang_in=previous_angular_velocity.clone(),
               sums_in=episode_sums.clone())
    for tag, terminal in (("t0", 0.0), ("t15", 1.5)):
        linear_acc = (v_forward_projected - previous_linear_velocity) / dt
        angular_acc = (base_velocities[:, 5] - previous_angular_velocity) / dt
        r1 = v_delta ** 2 + 5 * (omega_delta ** 2)
        r2 = linear_acc ** 2 + angular_acc ** 2
        r3 = torch.sum(wheel_torqs ** 2, dim=1)
        w1, w2, w3 = -25.0, -0.02, -0.09
        rdense = w1 * r1 + w2 * r2 + w3 * r3
        sparse_reward = torch.where((torch.abs(v_delta) < 0.01) & (torch.abs(omega_delta) < 0.01),
                                    torch.full_like(v_delta, 0.2), torch.zeros_like(v_delta))
        observed_reward = rdense + sparse_reward
        rew_buf = (1 - gamma2) * observed_reward + gamma2 * guiding_reward
        rew_buf = rew_buf + terminal * reset_buf * ~timeout_buf
        sums = episode_sums.clone()
        for k, val in enumerate((w1 * r1, w2 * r2, w3 * r3, rdense, sparse_reward, guiding_reward, observed_reward, rew_buf)):
            sums[:, k] += val
        out.update({f"lin_acc_{tag}": linear_acc, f"ang_acc_{tag}": angular_acc, f"rew_{tag}": rew_buf,
                    f"sums_out_{tag}": sums})
    # exercise the sparse bonus explicitly
    out.update(sparse_hit=((torch.abs(v_delta) < 0.01) & (torch.abs(omega_delta) < 0.01)).to(torch.uint8))

    # ---- KAMMA.py:701-710 forward projection ----
    forward_vec = torch.tensor([1.0, 0.0, 0.0]).repeat((n, 1))
    forward = quat_apply(q, forward_vec)
    forward_norm = forward / torch.norm(forward, dim=1, keepdim=True)
    v_global = base_velocities[:, 0:3]
    v_fwd = torch.sum(v_global * forward_norm, dim=1)
    out.update(v_fwd=v_fwd)

    # ---- KAMMA.py:833 observation + rl_task.py:99 clamp ----
    obs = torch.cat([desired_v.unsqueeze(1), desired_omega.unsqueeze(1), v_delta.unsqueeze(1), omega_delta.unsqueeze(1),
                     out["lin_acc_t0"].unsqueeze(1), out["ang_acc_t0"].unsqueeze(1)], dim=1)
    out.update(obs=torch.clamp(obs, -1000.0, 1000.0))
    return {k: v.numpy() for k, v in out.items()}


def refmath_reset(seed=1, n=513, cmd_x=(1.0, 2.0)):
    """KAMMA.py:417-460, 490-496, 569-575: pose / command part of reset_idx given the 8 draws.
    cmd_x = (0.0, 0.4) makes |commands[:, :2]| <= 0.25 common, so the zeroing at :458 is hit."""
    g = torch.Generator().manual_seed(seed)
    u = torch.rand(n, 8, generator=g, dtype=F)
    base_init_state = torch.tensor([0.0, 0.0, 0.1, 1.0, 0.0, 0.0, 0.0, 0, 0, 0, 0, 0, 0], dtype=F)
    cmd_y, cmd_yaw = (-0.5, 0.5), (1.0, 1.1)
    base_pos = base_init_state[0:3].repeat(n, 1)
    base_pos[:, 0:2] += torch_rand_float_from(-0.5, 0.5, u[:, 0:2])
    random_angle = torch_rand_float_from(0, 2 * torch.pi, u[:, 2])
    quat_z = torch.stack([torch.cos(random_angle / 2), torch.zeros_like(random_angle), torch.zeros_like(random_angle),
                          torch.sin(random_angle / 2)], dim=-1)
    base_quat = quat_mul(quat_z, base_init_state[3:7].repeat(n, 1))
    commands = torch.zeros(n, 4, dtype=F)
    commands[:, 0] = torch_rand_float_from(cmd_x[0], cmd_x[1], u[:, 3])
    commands[:, 1] = torch_rand_float_from(cmd_y[0], cmd_y[1], u[:, 4])
    commands[:, 3] = torch_rand_float_from(cmd_yaw[0], cmd_yaw[1], u[:, 5])
    commands *= (torch.norm(commands[:, :2], dim=1) > 0.25).unsqueeze(1)
    x_vel = torch_rand_float_from(cmd_x[0], cmd_x[1], u[:, 6])
    commands[:, 0] = torch.clamp(x_vel, min=0.0)  # max(x_vel, 0.0) per arena
    commands[:, 2] = 0.0
    return {"u": u.numpy(), "base_pos": base_pos.numpy(), "base_quat": base_quat.numpy(), "commands": commands.numpy(),
            "angle": random_angle.numpy()}


def refmath_extras(seed=2, n=64):
    """KAMMA.py:470-488: derived ratio sums and the means that go into extras['episode']."""
    g = torch.Generator().manual_seed(seed)
    sums = 10.0 * (torch.rand(n, 8, generator=g, dtype=F) - 0.7)
    gamma2 = torch.full((n,), 0.375, dtype=F)
    ids = torch.arange(0, n, 3)
    one, hundred, L = torch.tensor(1.0), torch.tensor(100.0), 10.0
    keys = {}
    for j, name in enumerate(["r1", "r2", "r3", "dense", "sparse"]):
        keys[name + "_over_final"] = L * (hundred * (one - gamma2) * sums[:, j] / sums[:, 7])
    keys["guiding_over_final"] = L * (hundred * gamma2 * sums[:, 5] / sums[:, 7])
    keys["observed_over_final"] = L * (hundred * (one - gamma2) * sums[:, 6] / sums[:, 7])
    out = {"sums": sums.numpy(), "gamma2": gamma2.numpy(), "ids": ids.numpy().astype(np.int32)}
    for j in range(8):
        out[f"mean_sum{j}"] = (torch.mean(sums[ids, j]) / L).numpy()
    for k, v in keys.items():
        out["mean_" + k] = (torch.mean(v[ids]) / L).numpy()
    return out


# ----------------------------------------------------------------------------------------
def refmath_gae(seed=3, T=24, n=41, discount_factor=0.99, lambda_coefficient=0.95):
    """Advantage estimation of the on-policy agents as torch loops over the rollout memory
    tensors [T, N, 1]: src/my_agents/ppo.py:345-358 (with the normalisation) and
    src/my_agents/ppoc.py:325-336. Python-double hyperparameters meet float32 tensors exactly
    as they do there."""
    g = torch.Generator().manual_seed(seed)
    rewards = torch.randn((T, n, 1), generator=g) * 0.5 - 0.3
    values = torch.randn((T, n, 1), generator=g)
    last_values = torch.randn((n, 1), generator=g)
    terminated = torch.rand((T, n, 1), generator=g) < 0.08
    terminated[-1, :5] = True  # a rollout that ends on a terminal step
    # --- ppo.py:345-358
    advantage = 0
    advantages = torch.zeros_like(rewards)
    not_dones = terminated.logical_not()
    for i in reversed(range(T)):
        next_values = values[i + 1] if i < T - 1 else last_values
        advantage = rewards[i] - values[i] + discount_factor * not_dones[i] * (next_values + lambda_coefficient * advantage)
        advantages[i] = advantage
    ppo_returns = advantages + values
    ppo_adv_raw = advantages.clone()
    ppo_adv = (advantages - advantages.mean()) / (advantages.std() + 1e-8)
    # --- ppoc.py:308-336
    term_f = terminated.float()
    next_vals = torch.zeros_like(values)
    next_vals[-1] = last_values
    adv2 = torch.zeros_like(rewards)
    gae = 0.0
    for t in reversed(range(T)):
        next_val = next_vals[t] if t == T - 1 else values[t + 1]
        delta = rewards[t] + discount_factor * (1.0 - term_f[t]) * next_val - values[t]
        gae = delta + discount_factor * lambda_coefficient * (1.0 - term_f[t]) * gae
        adv2[t] = gae
    ppoc_returns = adv2 + values
    return {"rewards": rewards.numpy(), "values": values.numpy(), "last_values": last_values.numpy(),
            "terminated": terminated.numpy(), "discount_factor": np.float64(discount_factor),
            "lambda_coefficient": np.float64(lambda_coefficient), "ppo_returns": ppo_returns.numpy(),
            "ppo_advantages_raw": ppo_adv_raw.numpy(), "ppo_advantages": ppo_adv.numpy(),
            "ppoc_returns": ppoc_returns.numpy(), "ppoc_advantages": adv2.numpy()}


# ----------------------------------------------------------------------------------------
def refmath_heights(seed=4, n=96):
    """get_heights (KAMMA.py:872-893) with the height-point grid of KAMMA.py:278-292 and
    quat_apply_yaw (KAMMA.py:895-899; normalize(x, eps=1e-9) = x / x.norm(p=2, dim=-1).clamp(min=eps)
    and quat_apply as published in omni.isaac.core.utils.torch), on a random int16 height field
    and random world poses, some of them outside the field (index clipping :886-887)."""
    g = torch.Generator().manual_seed(seed)
    rows, cols = 240, 180
    horizontal_scale, vertical_scale, border_size = 0.02, 0.006, 2
    height_samples = torch.randint(-300, 300, (rows, cols), generator=g, dtype=torch.int16)
    base_pos = torch.rand((n, 3), generator=g) * torch.tensor([3.0, 2.0, 1.0]) - torch.tensor([0.5, 0.5, 0.0])
    base_pos[:6] = torch.tensor([[-5.0, 1.0, 0], [30.0, 1.0, 0], [1.0, -9.0, 0], [1.0, 40.0, 0], [-2.1, -2.1, 0], [2.7, 1.5, 0]])
    base_quat = torch.nn.functional.normalize(torch.randn((n, 4), generator=g), dim=1)
    base_quat[::3, 1:3] *= 0.05  # mostly-upright robots
    base_quat = torch.nn.functional.normalize(base_quat, dim=1)
    # --- init_height_points :278-292
    y = 0.1 * torch.tensor([-5, -4, -3, -2, -1, 1, 2, 3, 4, 5])
    x = 0.1 * torch.tensor([-8, -7, -6, -5, -4, -3, -2, 2, 3, 4, 5, 6, 7, 8])
    grid_x, grid_y = torch.meshgrid(x, y, indexing="ij")
    num_height_points = grid_x.numel()
    height_points = torch.zeros(n, num_height_points, 3)
    height_points[:, :, 0] = grid_x.flatten()
    height_points[:, :, 1] = grid_y.flatten()
    # --- quat_apply_yaw :895-899
    quat = base_quat.repeat(1, num_height_points)
    quat_yaw = quat.clone().view(-1, 4)
    quat_yaw[:, 1:3] = 0.0
    quat_yaw = quat_yaw / quat_yaw.norm(p=2, dim=-1).clamp(min=1e-9, max=None).unsqueeze(-1)
    vec = height_points
    shape = vec.shape
    a, b = quat_yaw.reshape(-1, 4), vec.reshape(-1, 3)
    xyz = a[:, 1:]
    t = xyz.cross(b, dim=-1) * 2
    rotated = (b + a[:, :1] * t + xyz.cross(t, dim=-1)).view(shape)
    # --- get_heights :878-893
    points = rotated + (base_pos[:, 0:3]).unsqueeze(1)
    points += border_size
    points = (points / horizontal_scale).long()
    px = points[:, :, 0].view(-1)
    py = points[:, :, 1].view(-1)
    px = torch.clip(px, 0, height_samples.shape[0] - 2)
    py = torch.clip(py, 0, height_samples.shape[1] - 2)
    heights1 = height_samples[px, py]
    heights2 = height_samples[px + 1, py + 1]
    heights = torch.min(heights1, heights2)
    measured = heights.view(n, -1) * vertical_scale
    return {"height_samples": height_samples.numpy(), "base_pos": base_pos.numpy(), "base_quat": base_quat.numpy(),
            "horizontal_scale": np.float32(horizontal_scale), "vertical_scale": np.float32(vertical_scale),
            "border_size": np.float32(border_size), "measured_heights": measured.numpy()}


# ----------------------------------------------------------------------------------------
def refmath_curriculum(seed=6, n=384):
    """sample_velocity_command with self.curriculum (KAMMA.py:533-567) as post_physics_step calls it for
    every arena (KAMMA.py:719-725), plus the `fraction` line of reset_idx (KAMMA.py:498). The per-arena
    Python loop, the 0-dim tensor indexing (self.fraction[env_id]), math.sin on a tensor and the Python
    doubles (t, scale * t / max_episode_length_s) are kept as they are; torch.normal's draw is passed
    in as a standard normal z (normal(mean, std) = mean + std * z in float32), so any RNG can replay it
    and a negative std (where the reference's torch.normal raises) still has a value."""
    import math
    g = torch.Generator().manual_seed(seed)
    terrain_levels = torch.randint(0, 5, (n,), generator=g)
    r1_sums = -torch.rand(n, generator=g, dtype=F) * 14.0                      # episode sums of w1 * r1 <= 0
    r1_sums[::7] = torch.rand(len(r1_sums[::7]), generator=g, dtype=F) * 9.0   # positive values reach fraction > 0.5 (:548)
    threshold_high = 10.0                                                      # :269
    fraction = r1_sums / threshold_high                                        # :498
    z = torch.randn(n, generator=g, dtype=F)
    counters = torch.randint(1, 500000, (n,), generator=g)                     # episode_step_counter (:376, :684)
    dt = 10 * 0.01                                                             # :258-260 (Python double)
    max_episode_length_s = 10.0                                                # :261
    cmd0, cmd2 = torch.zeros(n, dtype=F), torch.zeros(n, dtype=F)
    for env_id in range(n):
        episode_step_counter = int(counters[env_id])
        normal = lambda mean, std: (torch.tensor(mean, dtype=F) + torch.as_tensor(std, dtype=F) * z[env_id]).item()
        if terrain_levels[env_id] == 0:
            sigma = 0.01 + 0.09 * fraction[env_id]                             # :538
            x_vel = normal(0.5, sigma)                                         # :539
            cmds = (0.0, max(x_vel, 0.0))                                      # :541
        elif terrain_levels[env_id] == 1:
            freq = 0.01 + 0.09 * fraction[env_id]                              # :546
            amp = 0.1                                                          # :547
            if fraction[env_id] > 0.5:
                amp = 0.1 + 0.4 * fraction[env_id]                             # :549
            t = float(episode_step_counter) * dt                               # :550
            x_vel = 0.5 + amp * math.sin(freq * t)                             # :551
            cmds = (max(x_vel, 0.0), 0.0)                                      # :552
        elif terrain_levels[env_id] == 2:
            t = float(episode_step_counter) * dt                               # :558
            scale = 0.5                                                        # :559
            Noise = 0.5 * fraction[env_id]                                     # :560
            x_vel = normal(0.0, Noise) + scale * t / max_episode_length_s      # :561
            cmds = (max(x_vel, 0.0), 0.0)                                      # :562
        else:
            x_vel = normal(0.5, 0.1)                                           # :566
            cmds = (max(x_vel, 0.0), 0.0)                                      # :567
        cmd0[env_id] = cmds[0]                                                 # :724 (assignment rounds to float32)
        cmd2[env_id] = cmds[1]                                                 # :725
    return {"terrain_levels": terrain_levels.numpy().astype(np.int32), "r1_sums": r1_sums.numpy(),
            "threshold_high": np.float32(threshold_high), "fraction": fraction.numpy(), "z": z.numpy(),
            "episode_step_counter": counters.numpy().astype(np.int32), "dt": np.float64(dt),
            "max_episode_length_s": np.float64(max_episode_length_s), "commands_0": cmd0.numpy(), "commands_2": cmd2.numpy()}


# ----------------------------------------------------------------------------------------
def evalsums_sequence(wheel_torqs, desired_v, v_delta, reset):
    """KAMMA_eval.py bookkeeping of its three extra episode sums over T steps of N arenas (also in KA_DDPG_1D/2D_eval):
    pre_physics_step :677-685 (Smoothness against old_torques, Desired velocity, then old_torques = the step's clipped
    torques), calculate_metrics :840 (Tracking error), reset_idx :470 (old_torques = zeros for ALL arenas, whichever
    arenas reset), :480-485 (rew_<key> = mean over the resetting arenas / max_episode_length_s, their sums zeroed).
    Inputs are [T, N(, 4)] tensors; returns the sums after every step [T, N, 3] (Smoothness, Tracking error, Desired
    velocity), old_torques after every step, and the extras means (NaN where no arena reset)."""
    T, n = desired_v.shape
    zeros = lambda: torch.zeros(n, dtype=F)  # noqa: E731
    sums = {"Smoothness": zeros(), "Tracking error": zeros(), "Desired velocity": zeros()}
    old_torques = torch.zeros(n, 4, dtype=F)                                      # :205
    max_episode_length_s = 10.0
    out_sums, out_old, out_rew = [], [], []
    for t in range(T):
        wt = wheel_torqs[t]
        diff = old_torques - wt                                                   # :677
        sums["Smoothness"] += torch.sum(diff ** 2, dim=1) / 10.0                  # :681
        sums["Desired velocity"] += desired_v[t] / 10.0                           # :683
        old_torques = wt.clone()                                                  # :685
        sums["Tracking error"] += torch.abs(v_delta[t]) / 10.0                    # :840
        env_ids = reset[t].nonzero(as_tuple=False).flatten()
        rew = torch.full((3,), float("nan"), dtype=F)
        if len(env_ids) > 0:                                                      # post_physics_step :691-693
            old_torques = torch.zeros(n, 4, dtype=F)                              # :470 (every arena)
            for k, key in enumerate(("Smoothness", "Tracking error", "Desired velocity")):
                rew[k] = torch.mean(sums[key][env_ids]) / max_episode_length_s    # :482-484
                sums[key][env_ids] = 0.0                                          # :485
        out_sums.append(torch.stack([sums["Smoothness"], sums["Tracking error"], sums["Desired velocity"]], 1).clone())
        out_old.append(old_torques.clone())
        out_rew.append(rew)
    return torch.stack(out_sums), torch.stack(out_old), torch.stack(out_rew)


def refmath_evalsums(seed=7, n=9, T=48):
    g = torch.Generator().manual_seed(seed)
    r = lambda *s, lo=-1.0, hi=1.0: (hi - lo) * torch.rand(*s, generator=g, dtype=F) + lo  # noqa: E731
    wheel_torqs = torch.clip(r(T, n, 4, lo=-5.0, hi=5.0), -4.0, 4.0)
    desired_v, v_delta = r(T, n, lo=0.0, hi=2.0), r(T, n, lo=-0.8, hi=0.8)
    reset = torch.zeros(T, n, dtype=torch.bool)
    reset[5, 2] = True; reset[6, 2] = True; reset[17, [0, 4, 8]] = True; reset[30] = True; reset[31, 7] = True; reset[44, 3] = True
    sums, old, rew = evalsums_sequence(wheel_torqs, desired_v, v_delta, reset)
    return {"wheel_torqs": wheel_torqs.numpy(), "desired_v": desired_v.numpy(), "v_delta": v_delta.numpy(),
            "reset": reset.numpy().astype(np.uint8), "sums": sums.numpy(), "old_torques": old.numpy(), "rew": rew.numpy()}


def trajectories():
    from oracle.binding import OracleEnv
    from tests.golden_replay import TRAJ_SPECS, run_spec
    return {name: run_spec(spec, lambda cfg: OracleEnv(cfg), to_numpy=lambda x: x) for name, spec in TRAJ_SPECS.items()}


def main():
    np.savez_compressed(os.path.join(HERE, "refmath_gae.npz"), **refmath_gae())
    np.savez_compressed(os.path.join(HERE, "refmath_heights.npz"), **refmath_heights())
    np.savez_compressed(os.path.join(HERE, "refmath_curriculum.npz"), **refmath_curriculum())
    np.savez_compressed(os.path.join(HERE, "refmath_evalsums.npz"), **refmath_evalsums())
    if "--gae-only" in sys.argv or "--unit-only" in sys.argv:
        return
    np.savez_compressed(os.path.join(HERE, "refmath_step.npz"), **refmath_step())
    np.savez_compressed(os.path.join(HERE, "refmath_reset.npz"), **refmath_reset())
    np.savez_compressed(os.path.join(HERE, "refmath_reset_small_cmd.npz"), **refmath_reset(seed=5, n=257, cmd_x=(0.0, 0.4)))
    np.savez_compressed(os.path.join(HERE, "refmath_extras.npz"), **refmath_extras())
    for name, d in trajectories().items():
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **d)
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)), "bytes")


if __name__ == "__main__":
    main()
