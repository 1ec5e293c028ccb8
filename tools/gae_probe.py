"""ois_gae throughput against the HBM roofline (run on the GPU box).
Algorithmic bytes per rollout entry: 4 (reward) + 4 (value) + 1 (terminated) read, 4 + 4 written
= 17 B; + 8 B (read + write of the advantage) for the normalisation pass."""
import json
import sys

import torch

sys.path.insert(0, ".")
from options_in_simulation_amd import compute_gae  # noqa: E402


def torch_loop(rewards, dones, values, last_values, g=0.99, l=0.95):  # the reference's structure (ppo.py:345-358)
    advantage = 0
    advantages = torch.zeros_like(rewards)
    not_dones = dones.logical_not()
    T = rewards.shape[0]
    for i in reversed(range(T)):
        nv = values[i + 1] if i < T - 1 else last_values
        advantage = rewards[i] - values[i] + g * not_dones[i] * (nv + l * advantage)
        advantages[i] = advantage
    returns = advantages + values
    advantages = (advantages - advantages.mean()) / (advantages.std() + 1e-8)
    return returns, advantages


def timeit(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters  # us


def direct(r, d, v, lv, normalize):
    """ois_gae itself with preallocated outputs (the wrapper's three allocations cost more host time than the kernel runs)"""
    import ctypes as C
    from options_in_simulation_amd import _native
    L = _native.load()
    T, N = r.shape[0], r.shape[1]
    ret, adv = torch.empty_like(r), torch.empty_like(r)
    nb = L.ois_gae_scratch_bytes(N)
    scratch = torch.empty(nb // 8, dtype=torch.float64, device=r.device)
    d8 = d.view(torch.uint8)
    stream = torch.cuda.current_stream().cuda_stream
    return lambda: L.ois_gae(0, T, N, r.data_ptr(), d8.data_ptr(), v.data_ptr(), lv.data_ptr(), 0.99, 0.95, int(normalize),
                             ret.data_ptr(), adv.data_ptr(), scratch.data_ptr(), nb, stream)


out = []
for T, N in [(16, 65536), (20, 65536), (32, 65536), (96, 65536), (1024, 65536), (16, 131072), (16, 262144), (16, 524288)]:
    r = torch.randn((T, N, 1), device="cuda"); v = torch.randn((T, N, 1), device="cuda")
    lv = torch.randn((N, 1), device="cuda"); d = torch.rand((T, N, 1), device="cuda") < 0.05
    us_raw = timeit(direct(r, d, v, lv, False), 200)
    us_norm = timeit(direct(r, d, v, lv, True), 200)
    us_wrapped = timeit(lambda: compute_gae(r, d, v, lv), 50)
    us_ref = timeit(lambda: torch_loop(r, d, v, lv), 3 if T > 100 else 10)
    n = T * N
    out.append({"T": T, "N": N, "us_scan": round(us_raw, 1), "GBps_scan": round(17 * n / us_raw / 1e3, 1),
                "us_scan_plus_normalize": round(us_norm, 1), "launches_normalized": 2 if (N + 255) // 256 <= 2048 and N <= 262144 else 3,  # ois_gae.hip: scan blocks of 256 up to 262 144 arenas, fold up to 2048 partials
                "GBps_total": round(25 * n / us_norm / 1e3, 1),
                "us_through_the_python_wrapper": round(us_wrapped, 1),
                "us_torch_loop": round(us_ref, 1), "speedup_vs_torch_loop": round(us_ref / us_norm, 1)})
    print(json.dumps(out[-1]), flush=True)
json.dump(out, open("gpurun_out/gae_probe.json", "w"), indent=1)
