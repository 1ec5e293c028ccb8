"""the reference's task math around the physics — controller,
gamma, switch, clip, is_done, calculate_metrics, observation, projections (KAMMA.py:604-656,
674-833), one torch op per reference op as the reference runs it — on the same MI355X at 65 536
arenas, WITHOUT the physics (PhysX there) and without reset_idx's per-arena Python loop. Shows what
the launch count alone costs next to the fused step (which includes 20 physics substeps)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from options_in_simulation_amd import _abi, build_config
from options_in_simulation_amd.vecenv import NativeEnv

N = 65536
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
r = lambda *s: torch.rand(*s, device=dev, generator=g)  # noqa: E731
commands = torch.stack([1 + r(N), r(N) - 0.5, torch.zeros(N, device=dev), 1 + 0.1 * r(N)], 1)
v_forward_projected, base_velocities, actions = r(N) * 2, r(N, 6) - 0.5, r(N, 4)
base_quat = torch.nn.functional.normalize(torch.tensor([[1.0, 0.0, 0.0, 0.3]], device=dev) + 0.05 * r(N, 4), dim=1)
base_pos = r(N, 3)
sim_steps = torch.zeros(N, dtype=torch.long, device=dev)
episode_buf = torch.zeros(N, dtype=torch.long, device=dev)
prev_lin, prev_ang = torch.zeros(N, device=dev), torch.zeros(N, device=dev)
episode_sums = {k: torch.zeros(N, device=dev) for k in range(8)}
gravity_vec = torch.tensor([0.0, 0.0, -1.0], device=dev).repeat((N, 1))
forward_vec = torch.tensor([1.0, 0.0, 0.0], device=dev).repeat((N, 1))
bounds = torch.tensor([-50.0, 50.0, -50.0, 50.0], device=dev)


def quat_apply(a, b):
    xyz = a[:, 1:]
    t = xyz.cross(b, dim=-1) * 2
    return b + a[:, :1] * t + xyz.cross(t, dim=-1)


def task_math_step():
    global prev_lin, prev_ang
    Kp, Kpo, m, J, dt, scale = 0.035, 0.665, 25.0, 1.05, 0.1, 3.0
    desired_v, desired_omega = commands[:, 0], commands[:, 2]
    current_v, current_omega = v_forward_projected, base_velocities[:, 5]
    v_delta, omega_delta = desired_v - current_v, desired_omega - current_omega
    ac_left = Kp * (m * (v_delta / dt)) - Kpo * (J * (omega_delta / dt))
    ac_right = Kp * (m * (v_delta / dt)) + Kpo * (J * (omega_delta / dt))
    criteria = torch.stack([ac_left, ac_left, ac_right, ac_right], dim=1)
    gamma = torch.clamp(1.0 - (sim_steps.float() / 400000.0), min=0.0)
    mask = (torch.rand(N, 1, device=dev) < gamma.view(-1, 1)).float()
    execution = mask * criteria + (1 - mask) * (actions * scale)
    wheel_torqs = torch.clip(execution, -4.0, 4.0)
    guiding = -torch.norm(wheel_torqs - criteria, dim=1)
    # (PhysX: 20 world steps here)
    episode_buf.add_(1); sim_steps.add_(1)
    timeout = torch.where(episode_buf >= 99, torch.ones_like(episode_buf), torch.zeros_like(episode_buf))
    fallen = quat_apply(base_quat, gravity_vec)[:, 2] > 0.0
    reset = torch.where(timeout.bool(), torch.ones_like(fallen), fallen)
    oob = (base_pos[:, 0] < bounds[0]) | (base_pos[:, 0] > bounds[1]) | (base_pos[:, 1] < bounds[2]) | (base_pos[:, 1] > bounds[3])
    reset = torch.where(oob, torch.ones_like(reset), reset)
    lin_acc = (v_forward_projected - prev_lin) / dt
    ang_acc = (base_velocities[:, 5] - prev_ang) / dt
    prev_lin, prev_ang = v_forward_projected.clone(), base_velocities[:, 5].clone()
    r1 = v_delta ** 2 + 5 * (omega_delta ** 2)
    r2 = lin_acc ** 2 + ang_acc ** 2
    r3 = torch.sum(wheel_torqs ** 2, dim=1)
    rdense = -25.0 * r1 + -0.02 * r2 + -0.09 * r3
    sparse = torch.where((torch.abs(v_delta) < 0.01) & (torch.abs(omega_delta) < 0.01), torch.full_like(v_delta, 0.2), torch.zeros_like(v_delta))
    observed = rdense + sparse
    rew = (1 - gamma) * observed + gamma * guiding
    rew = rew + 0.0 * reset * ~timeout
    for k, val in enumerate((-25.0 * r1, -0.02 * r2, -0.09 * r3, rdense, sparse, guiding, observed, rew)):
        episode_sums[k] += val
    obs = torch.cat([desired_v.unsqueeze(1), desired_omega.unsqueeze(1), v_delta.unsqueeze(1), omega_delta.unsqueeze(1),
                     lin_acc.unsqueeze(1), ang_acc.unsqueeze(1)], dim=1)
    obs = torch.clamp(obs, -1000.0, 1000.0)
    forward = quat_apply(base_quat, forward_vec)
    forward = forward / torch.norm(forward, dim=1, keepdim=True)
    v_fwd = torch.sum(base_velocities[:, 0:3] * forward, dim=1)
    return obs, rew, reset, v_fwd


def timeit(fn, n):
    for _ in range(10):
        fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


t_ref = timeit(task_math_step, 200)
env = NativeEnv(build_config(variant="KAMMA_medium", num_envs=N, seed=1), dev)
env.reset(1)
t_fused = timeit(lambda: env.step(actions), 200)
out = {"arenas": N, "reference_style_task_math_us_per_step": round(t_ref, 1), "fused_step_incl_physics_us_per_step": round(t_fused, 1),
       "note": "task math only, no physics, no reset loop, unfused torch ops as the reference issues them; fused = ois_step incl. 20 physics substeps"}
print(json.dumps(out))
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/r1_reference_style_probe.json", "w"), indent=1)
