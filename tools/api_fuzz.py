"""Random interleavings of the C-ABI's calls, kernel vs CPU oracle (GPU box): per sequence an environment of a random size and
variant, then OPS random operations — ois_step, ois_rollout (EXTERNAL / RANDOM, 1-5 steps), ois_reset, ois_push_robots,
ois_randomize_params, ois_save_state / ois_load_state round trips, attach / detach of a height field (either footprint stride),
attach / detach of a transition memory (ois_replay_*: every tensor, the cursor and a sampled mini-batch compared) —
and, after every one of them, a random subset of the reads a trainer makes: the published logs (ois_logs_fetch through
NativeEnv.host_logs), ois_episode_info, ois_env0_diag, ois_stats_fetch (with and without reset), ois_outputs_copy, ois_height_scan. Every read and
every state buffer must equal the oracle's, bit for bit. What it is after: the host-side bookkeeping around the kernels (which
reduction is current, what a cached record belongs to, which shell a changed environment launches).
    python tools/api_fuzz.py [SEQUENCES] [OPS]"""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from options_in_simulation_amd import _abi, build_config  # noqa: E402
from options_in_simulation_amd import terrain as tg  # noqa: E402
from options_in_simulation_amd.config import VARIANTS  # noqa: E402
from options_in_simulation_amd.vecenv import NativeEnv  # noqa: E402
from oracle.binding import OracleEnv  # noqa: E402

SEQ = int(sys.argv[1]) if len(sys.argv) > 1 else 200
OPS = int(sys.argv[2]) if len(sys.argv) > 2 else 60
STATE = [_abi.BUF_OBS, _abi.BUF_REW, _abi.BUF_DONE, _abi.BUF_TIMEOUT, _abi.BUF_WHEEL_TORQS, _abi.BUF_COMMANDS, _abi.BUF_POSE, _abi.BUF_QUAT,
         _abi.BUF_LINVEL, _abi.BUF_ANGVEL, _abi.BUF_WHEEL_VEL, _abi.BUF_COUNTERS, _abi.BUF_SUMS_A, _abi.BUF_SUMS_B, _abi.BUF_SUMS_C, _abi.BUF_LAMBDA_N,
         _abi.BUF_LAMBDA_L, _abi.BUF_LAMBDA_T, _abi.BUF_ENV0_DIAG, _abi.BUF_PARAMS, _abi.BUF_ENV_ORIGINS, _abi.BUF_TERRAIN_IDX]
STAT = ("episodes", "return_sum", "return_sqsum", "return_min", "return_max", "length_sum", "steps")
trace = []


def fail(msg):
    print("DIFFERENCE", msg)
    print("   sequence so far:", " | ".join(trace[-25:]))
    sys.exit(1)


def same(dev, orc, where):
    torch.cuda.synchronize()
    for bid in STATE:
        got = dev.buffer(bid).cpu().numpy()
        if got.dtype == np.bool_:
            got = got.astype(np.uint8)
        if not np.array_equal(got, orc.buffer(bid), equal_nan=True):
            fail(f"{where}: buffer {bid} at {np.argwhere(got != orc.buffer(bid))[:3].tolist()}")


variants = sorted(VARIANTS)
t0 = time.time()
nops = 0
for seq in range(SEQ):
    rng = np.random.default_rng(77000 + seq)
    # sizes either side of the shells' and the logging reductions' thresholds (lane pair <= 32 768, one workgroup <= 131 072 arenas)
    N = int(rng.choice([1, 2, 63, 65, 300, 1000, 4097, 20000, 40000, 70000, 140000], p=[.12, .08, .1, .1, .14, .14, .1, .1, .05, .04, .03]))
    variant = variants[rng.integers(len(variants))]
    os.environ["OIS_STEP_WAVES"] = str(rng.integers(1, 3))
    os.environ.pop("OIS_LANES_PER_ARENA", None)   # unset = by size: a lane pair per arena up to 32 768 arenas, one lane on a height field
    if rng.random() < 0.5:
        os.environ["OIS_LANES_PER_ARENA"] = "1"
    cfg = build_config(variant=variant, num_envs=N, seed=int(rng.integers(1 << 30)), max_sim_steps=float(rng.uniform(30, 200)))
    cfg.max_episode_length = int(rng.integers(4, 30))   # resets all the time: the records change often
    dev, orc = NativeEnv(cfg, "cuda:0"), OracleEnv(cfg, threads=16)
    T = tg.Terrain({"numLevels": 2, "numTerrains": 2, "mapLength": 8.0, "mapWidth": 8.0}, N, seed=seq)
    lt = T.sample_levels_types(N, rng=np.random.RandomState(seq))
    seed = int(rng.integers(1 << 30))
    dev.reset(seed); orc.reset(seed)
    trace[:] = [f"seq {seq}: {variant} N={N} waves={os.environ['OIS_STEP_WAVES']} lanes={os.environ.get('OIS_LANES_PER_ARENA', 'by size')}", "reset"]
    lo = -1.0 if variant.startswith("KA_DDPG") else 0.0
    stepped = False
    mem_dev = mem_orc = None
    for op in range(OPS):
        r = rng.random()
        if r < 0.45:
            a = rng.uniform(lo, 1.0, size=(N, cfg.action_dim)).astype(np.float32)
            dev.step(torch.from_numpy(a).cuda()); orc.step(a); trace.append("step"); stepped = True
        elif r < 0.60:
            k = int(rng.integers(1, 6))
            if cfg.eval_sums:
                k = 1
            if rng.random() < 0.5:
                acts = rng.uniform(lo, 1.0, size=(k, N, cfg.action_dim)).astype(np.float32)
                dev.rollout(k, _abi.POLICY_EXTERNAL, torch.from_numpy(acts).cuda()); orc.rollout(k, _abi.POLICY_EXTERNAL, acts)
                trace.append(f"rollout({k}, external)")
            else:
                dev.rollout(k, _abi.POLICY_RANDOM); orc.rollout(k, _abi.POLICY_RANDOM); trace.append(f"rollout({k}, random)")
            stepped = True
        elif r < 0.65:
            seed = int(rng.integers(1 << 30))
            dev.reset(seed); orc.reset(seed); trace.append("reset"); stepped = False
        elif r < 0.70:
            dev.push_robots(); orc.push_robots(); trace.append("push_robots")
        elif r < 0.75:
            sl = [-0.2, 0.0, 0.1] if rng.random() < 0.6 else []
            dev.randomize_params(0.4, 1.1, sl); orc.randomize_params(0.4, 1.1, sl); trace.append(f"randomize_params(slopes={len(sl)})")
        elif r < 0.82:
            blob = dev.save_state()
            snap = {k: v.clone() for k, v in mem_dev.items()} if mem_dev is not None else None   # the memory tensors are the caller's
            a = rng.uniform(lo, 1.0, size=(N, cfg.action_dim)).astype(np.float32)
            dev.step(torch.from_numpy(a).cuda())            # wander off ...
            dev.load_state(blob); trace.append("save, step, load")   # ... and come back: the oracle never left
            if snap is not None:
                for k in snap:
                    mem_dev[k].copy_(snap[k])
        elif r < 0.88:
            if getattr(dev, "_terrain", None) is None:
                kw = {"footprint_stride": int(rng.integers(1, 3))}
                if VARIANTS[variant].get("cmd_sampler") != _abi.CMD_EVAL and rng.random() < 0.5:
                    # the level curriculum: levels move at the resets, the commands follow the per-level generators, and the logging
                    # reductions cannot take the level mean from the host any more (two launches)
                    kw.update(curriculum=True, threshold_low=-60.0, threshold_high=-20.0, max_level=1)
                dev.attach_terrain(T, lt, **kw); orc.attach_terrain(T, lt, **kw); trace.append(f"attach_terrain({kw})")
            else:
                dev.detach_terrain(); orc.detach_terrain(); trace.append("detach_terrain")
            seed = int(rng.integers(1 << 30))
            dev.reset(seed); orc.reset(seed); trace.append("reset"); stepped = False
        elif r < 0.93:
            if mem_dev is None:
                M = int(rng.integers(1, 10))
                mem_dev, mem_orc = dev.attach_replay(M), orc.attach_replay(M); trace.append(f"attach_replay({M})")
            else:
                dev.detach_replay(); orc.attach_replay(0); mem_dev = mem_orc = None; trace.append("detach_replay")
        else:
            trace.append("(reads only)")
        nops += 1
        same(dev, orc, f"seq {seq} op {op}")
        if mem_dev is not None:   # the FIFO transition memory the step kernel writes (train_kaddpg_4d_nofifo_random_RLIL.py:147-156)
            torch.cuda.synchronize()
            if dev.replay_cursor() != orc.replay_cursor():
                fail(f"seq {seq} op {op}: memory cursor {dev.replay_cursor()} vs {orc.replay_cursor()}")
            for k in mem_dev:
                got = mem_dev[k].cpu().numpy()
                if not np.array_equal(got.astype(mem_orc[k].dtype), mem_orc[k], equal_nan=True):
                    fail(f"seq {seq} op {op}: transition memory '{k}'")
            if rng.random() < 0.3:
                rows = mem_orc["rewards"].shape[0] * N
                idx = rng.integers(0, rows, size=int(rng.integers(1, 300)))
                batch = dev.replay_sample(torch.from_numpy(idx))
                torch.cuda.synchronize()
                for k in batch:
                    want = mem_orc[k].reshape(rows, -1)[idx]
                    if not np.array_equal(batch[k].cpu().numpy().astype(want.dtype), want, equal_nan=True):
                        fail(f"seq {seq} op {op}: sampled '{k}'")
        # the reads of a trainer, in a random subset and order
        reads = [x for x in ("logs", "episode_info", "diag", "stats", "stats_reset", "outputs", "logs", "height_scan") if rng.random() < 0.4]
        rng.shuffle(reads)
        for what in reads:
            trace.append("read " + what)
            if what == "logs" and stepped:
                ei, diag = dev.host_logs()[:2]
                if not np.array_equal(ei, orc.episode_info(), equal_nan=True):
                    fail(f"seq {seq} op {op}: published episode_info {ei[:6]} vs {orc.episode_info()[:6]}")
                if not np.array_equal(diag, orc.buffer(_abi.BUF_ENV0_DIAG), equal_nan=True):
                    fail(f"seq {seq} op {op}: published env-0 diagnostics")
            elif what == "episode_info":
                if not np.array_equal(dev.episode_info().cpu().numpy(), orc.episode_info(), equal_nan=True):
                    fail(f"seq {seq} op {op}: ois_episode_info")
            elif what == "diag":
                if not np.array_equal(dev.env0_diag(direct=bool(rng.integers(2))), orc.buffer(_abi.BUF_ENV0_DIAG), equal_nan=True):
                    fail(f"seq {seq} op {op}: ois_env0_diag")
            elif what in ("stats", "stats_reset"):
                ra = what == "stats_reset"
                a_, b_ = dev.stats(reset_after=ra), orc.stats(reset_after=ra)
                for f in STAT:
                    x, y = getattr(a_, f), getattr(b_, f)
                    if not (x == y or (np.isnan(x) and np.isnan(y)) or abs(x - y) <= 1e-9 * max(1.0, abs(y))):
                        fail(f"seq {seq} op {op}: statistics {f}: {x} vs {y}")
            elif what == "height_scan":
                if getattr(dev, "_terrain", None) is not None and not np.array_equal(dev.height_scan().cpu().numpy(), orc.height_scan()):
                    fail(f"seq {seq} op {op}: ois_height_scan")
            elif what == "outputs":
                o, rw, te, tr = dev.outputs_copy()
                torch.cuda.synchronize()
                if not (np.array_equal(o.cpu().numpy(), orc.obs, equal_nan=True) and np.array_equal(rw.cpu().numpy().reshape(-1), orc.rew.reshape(-1), equal_nan=True)
                        and np.array_equal(te.cpu().numpy().reshape(-1).astype(np.uint8), orc.done.reshape(-1)) and not bool(tr.any())):
                    fail(f"seq {seq} op {op}: ois_outputs_copy")
    dev.close(); orc.close()
    if seq % 10 == 9:
        print(f"ok {seq + 1} sequences, {nops} operations", flush=True)
print(f"{SEQ} sequences x {OPS} operations (+ reads after each), all equal to the oracle, {time.time() - t0:.0f} s")
