#!/usr/bin/env python3
"""Headline benchmark: env-steps/sec at 65 536 parallel arenas per GPU (BASELINE.json).

Workload (BASELINE.json configs[2]): 65 536 arenas per GPU, flat terrain, KAMMA_medium task,
scripted PPO-style actor (the train scripts' MLP 6->512->512->128->4, ELU, Sigmoid head,
train_kamma_4d_nofifo_random_RLIL.py:21-36; fixed-seed random weights, fp32 PyTorch-ROCm
inference) in the loop: obs -> actor -> ois_step, all on one stream, no host sync.

One "step" = one policy step of all arenas of this rank. N>1: one process per GPU
(torch.distributed / RCCL), arenas sharded by rank (weak scaling: 65 536 per GPU; global
arena ids keep every per-arena stream identical to a 1-GPU run — the benchmark's KAMMA_medium task has
no per-environment quantity; the eval / level-curriculum variants have two), ONE
all-gather of the episode-return statistics per logging interval (800 steps,
train_kamma_4d_nofifo_random_RLIL.py:176) — the path's only exchange.

`python bench.py --gpus N` with N > 1 and no torch.distributed environment launches its own N
ranks (a child `python -m torch.distributed.run`, started before anything in this process touches
the GPU; the parent never initialises it and never execs); under the driver's own
torch.distributed.run launch it is simply one of the ranks.

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
  roofline      of the step kernel, against the ceiling that binds it (fp32 VALU issue; the HBM
                figure on the algorithmic 352 B/env-step of SURVEY.md section 8d sits beside it).
                The kernel time is measured live: start / stop events on the kernel's own dispatch
                packets over the timed region (ois_profile_*), i.e. what rocprofv3 --kernel-trace
                reports for the same launches. FLOPs three ways: executed by the launched instantiation
                (ISA census), algorithmic (docs/SPEC.md 3.11), and the issue-slot utilisation from the
                committed SQ counters — each quoted only when its artefact was taken on this checkout's
                kernel sources.
  cpu_baseline  the CPU oracle (oracle/, test infrastructure) timed on this host's cores on a
                bounded sample of the same workload (rank 0, N=1 only).
"""
import argparse
import contextlib
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALGO_BYTES_PER_ENV_STEP = 352.0
HBM_PEAK_GBS = 8000.0             # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
# The ceiling that binds (SURVEY.md §8(d) asks to say which): fp32 vector peak, and the arithmetic
# one env-step performs, counted from the ISA of the packed step kernel's substep loop
# (tools/isa_census.py: 614 executed VALU instructions, 344 of them packed = 1 440 FLOP per
# arena-substep with fma = 2; x 20 substeps; pre/post-physics not counted)
FP32_VECTOR_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md
FLOP_PER_ENV_STEP = 20 * 1440.0   # fallback only: the general packed kernel (profiles/*_isa_census.json keys the launched one)
ARENAS_PER_GPU = 65536
ROUND = "r4"                      # the profiles/<round>_* artefacts this checkout quotes
KERNEL_SOURCES = ("ois_kernels.hip", "ois_device.h", "ois_physics.h", "ois_physics_pk.h", "Makefile")


def kernel_source_hash():
    """sha256 over the step kernel's sources: ties profiles/*_pmc_traffic.json to the kernel at HEAD"""
    h = hashlib.sha256()
    for name in KERNEL_SOURCES:
        with open(os.path.join(ROOT, "options_in_simulation_amd", "csrc", name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def build_actor(torch, device):
    """train_kamma_4d_nofifo_random_RLIL.py:21-36: 6 -> 512 -> 512 -> 128 -> 4, ELU, Sigmoid"""
    nn = torch.nn
    torch.manual_seed(0)
    # inference only: the ELUs work in place (same values; 487 -> 471 us per 65 536-row pass)
    net = nn.Sequential(nn.Linear(6, 512), nn.ELU(inplace=True), nn.Linear(512, 512), nn.ELU(inplace=True),
                        nn.Linear(512, 128), nn.ELU(inplace=True), nn.Linear(128, 4), nn.Sigmoid())
    return net.to(device).eval()


def usable_cores():
    """threads the CPU baseline may use: the cgroup CPU quota when there is one, otherwise the
    affinity mask capped at the GPU box's per-GPU CPU share (16)."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, int(os.environ.get("OIS_CPU_THREADS", "16"))))


def measured_copy_gbs(torch, device, mb=1024, reps=10):
    """what a plain device copy achieves on this GPU (read + write bytes / time), the practical
    ceiling SURVEY.md section 8(d) asks to quote beside the 8 TB/s spec peak (tools/copy_bw.hip is the
    same measurement with an in-repo kernel: profiles/r1_copy_bw.txt)."""
    n = (mb << 20) // 4
    src = torch.ones(n, dtype=torch.float32, device=device)
    dst = torch.empty_like(src)
    dst.copy_(src)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        dst.copy_(src)
    e1.record()
    torch.cuda.synchronize(device)
    return 2.0 * n * 4 * reps / (e0.elapsed_time(e1) * 1e-3) / 1e9


def algorithmic_flop_per_env_step():
    """operations of the MODEL (docs/SPEC.md section 3.11, tools/spec_flops.py) — independent of any kernel's instructions"""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import spec_flops
    return float(spec_flops.flop_per_env_step())


def executed_flop_per_env_step(kernel_id, substeps, profile=None):
    """fp32 operations the LAUNCHED step-kernel instantiation executes per arena and env-step, from the committed ISA census
    (tools/isa_census.py --json; quoted only when taken on this checkout's kernel sources). Returns (flop, valu, provenance)."""
    profile = profile or f"{ROUND}_isa_census.json"
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", profile)))
    except (OSError, ValueError):
        return FLOP_PER_ENV_STEP, None, {"file": "profiles/" + profile, "status": "missing: fallback constant of the general packed kernel"}
    prov = {"file": "profiles/" + profile, "kernel_src_sha256": d.get("kernel_src_sha256"), "kernel": kernel_id}
    k = d.get("kernels", {}).get(kernel_id)
    if d.get("kernel_src_sha256") != kernel_source_hash() or k is None:
        prov["status"] = "not quoted: the kernel sources changed since the census (or the instantiation is not in it); fallback constant"
        return FLOP_PER_ENV_STEP, None, prov
    lanes = 2 if kernel_id.startswith("step_kernel<2") else 1
    prov["status"] = "quoted: executed path of the substep loop x substeps x lanes per arena (pre / post-physics not counted)"
    per = float(k.get("loop_substeps", 1))  # substeps one iteration of the census's loop holds (2 with footprint stride 2)
    return float(k["substep_flop_per_lane"] * lanes * substeps) / per, float(k["substep_valu"] * substeps) / per, prov


def pmc_sq(n_arenas, profile=None):
    """issue-slot figures of single ois_step launches in the actor loop from the committed SQ counter run
    (tools/profile_sq.sh; quoted under the same rule as `traffic`)"""
    profile = profile or f"{ROUND}_pmc_sq.json"
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", profile)))
    except (OSError, ValueError):
        return None, {"file": "profiles/" + profile, "status": "missing"}
    prov = {"file": "profiles/" + profile, "kernel_src_sha256": d.get("kernel_src_sha256"), "mode": d.get("mode")}
    if int(d.get("arenas", -1)) != int(n_arenas) or d.get("kernel_src_sha256") != kernel_source_hash():
        prov["status"] = "not quoted: profile taken at another arena count or on other kernel sources"
        return None, prov
    prov["status"] = "quoted: rocprofv3 --pmc, one pass per counter, SQ_* in quad-cycles"
    return {"valu_instructions_per_wave_per_env_step": d["valu_instr_per_wave_per_step"],
            "quad_cycles_per_valu_instruction": d["quad_cycles_per_valu_instr"],
            "valu_active_fraction_of_wave_life": d["valu_active_fraction_of_wave_life"], "clock_GHz": d["clock_GHz_est"],
            "issue_limited_us": d["issue_limited_ns"] * 1e-3, "dispatch_us": d["dispatch_ns"] * 1e-3,
            "issue_slot_utilisation": d["issue_slot_utilisation"]}, prov


def pmc_traffic(n_arenas, profile=None):
    """HBM bytes per step_kernel launch from the committed PMC run (bench.py cannot collect counters
    on itself). Quoted only when the profile was taken at this arena count AND on the kernel sources
    of this checkout (sha256 recorded by tools/profile_round.sh); returns (bytes | None, provenance)."""
    profile = profile or f"{ROUND}_pmc_traffic.json"
    path = os.path.join(ROOT, "profiles", profile)
    try:
        d = json.load(open(path))
    except (OSError, ValueError):
        return None, {"file": "profiles/" + profile, "status": "missing"}
    prov = {"file": "profiles/" + profile, "profile_head": d.get("git_head"), "kernel_src_sha256": d.get("kernel_src_sha256")}
    if int(d.get("arenas", -1)) != int(n_arenas):
        prov["status"] = f"not quoted: profile taken at {d.get('arenas')} arenas"
        return None, prov
    if d.get("kernel_src_sha256") != kernel_source_hash():
        prov["status"] = "not quoted: the kernel sources changed since the profile was taken"
        return None, prov
    prov["status"] = "quoted: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, FETCH_SIZE x2 per the gfx950 correction"
    return d["hbm_bytes_per_launch"], prov


def cpu_baseline(n_arenas, seconds_target=12.0):
    """time the oracle on the host cores: same task, same actor-free step (random actions)."""
    from options_in_simulation_amd import _abi, build_config
    from oracle.binding import OracleEnv
    cores = usable_cores()
    n = min(n_arenas, 4096)
    env = OracleEnv(build_config(variant="KAMMA_medium", num_envs=n, seed=42), threads=cores)
    env.reset(42)
    env.rollout(2, _abi.POLICY_RANDOM)  # warm the thread pool
    t0 = time.perf_counter()
    env.rollout(4, _abi.POLICY_RANDOM)
    per_step = (time.perf_counter() - t0) / 4
    steps = int(max(8, min(50000, seconds_target / max(per_step, 1e-6))))
    t0 = time.perf_counter()
    env.rollout(steps, _abi.POLICY_RANDOM)
    dt = time.perf_counter() - t0
    env.close()
    # the same oracle on ONE thread (512 arenas, ~3 s), rate per core
    n1 = min(n_arenas, 512)
    one = OracleEnv(build_config(variant="KAMMA_medium", num_envs=n1, seed=42), threads=1)
    one.reset(42)
    one.rollout(2, _abi.POLICY_RANDOM)
    t0 = time.perf_counter()
    one.rollout(4, _abi.POLICY_RANDOM)
    steps1 = int(max(4, min(5000, 3.0 / max((time.perf_counter() - t0) / 4, 1e-6))))
    t0 = time.perf_counter()
    one.rollout(steps1, _abi.POLICY_RANDOM)
    dt1 = time.perf_counter() - t0
    one.close()
    return {"value": n * steps / dt, "unit": "env-steps/s", "cores": cores, "kind": "port",
            "sample": f"{n} arenas x {steps} policy steps of the same KAMMA_medium task (in-oracle random actions, "
                      f"no actor), OpenMP over arenas, {dt:.1f} s",
            "single_thread": {"value": n1 * steps1 / dt1, "cores": 1,
                              "sample": f"{n1} arenas x {steps1} policy steps, {dt1:.1f} s"}}


def free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def spawn_command(gpus, argv, port):
    """the rank launch of a bare `python bench.py --gpus N`: the command line the driver itself uses"""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}",
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *argv]


def maybe_self_launch(args, argv):
    """--gpus N > 1 outside a torch.distributed launch: start the N ranks as a child process and
    pass its exit code on. Nothing before this point imports torch or touches the GPU, and the
    parent only waits: it never initialises the GPU and never execs."""
    if args.gpus <= 1 or "RANK" in os.environ or "LOCAL_RANK" in os.environ:
        return
    env = dict(os.environ)
    # the host driver of this pool only supports dmabuf IPC: without it RCCL's cross-process buffer sharing fails with
    # hipIpcGetMemHandle: invalid argument (already exported on the pool's boxes; kept for a shell that lost it)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # torch.distributed.run sets OMP_NUM_THREADS=1 per rank when it is unset and warns about it; the ranks' host work is a
    # few ctypes calls per step, so a small fixed number is enough and leaves the box's 16 cores per GPU unclaimed
    env.setdefault("OMP_NUM_THREADS", "4")
    sys.exit(subprocess.call(spawn_command(args.gpus, argv, free_port()), env=env))


def parse_args(argv):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--arenas", type=int, default=ARENAS_PER_GPU, help="arenas per GPU")
    ap.add_argument("--log-interval", type=int, default=800)
    ap.add_argument("--prewarm", type=int, default=40,
                    help="iterations of the same loop on a scratch environment before the W warm-up steps (untimed; 0 = off)")
    ap.add_argument("--shards", type=int, default=1, metavar="K",
                    help="A/B: the rank's arenas as K environments (arena_offset i * N / K) on K HIP streams, the actor on each "
                         "shard's rows — the env step of one shard can overlap the actor of another; same arenas, same results")
    ap.add_argument("--no-actor", action="store_true", help="env-only loop (fixed random actions)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the env-only / fused side measurements after the timed region")
    ap.add_argument("--fused", type=int, default=0, metavar="K",
                    help="env-only, K policy steps per launch with the in-kernel random-torque policy (ois_rollout; "
                         "BASELINE.json configs[1]'s policy at this arena count): one timed 'step' = one launch of K steps")
    ap.add_argument("--variant", default="KAMMA_medium")
    ap.add_argument("--domain-rand", action="store_true",
                    help="BASELINE.json configs[3]: per-arena friction U(0.5, 1.0) and plane slope from the generator's set")
    ap.add_argument("--terrain", action="store_true",
                    help="configs[3] on the reference's full terrain family: 6 x 2 random tiles, wheel contact on the height field")
    ap.add_argument("--footprint-stride", type=int, default=1, choices=(1, 2),
                    help="with --terrain: ois_terrain.footprint_stride (2 = wheel footprints sampled every second substep, docs/SPEC.md 3.8)")
    return ap.parse_args(argv)


def side_measurements(torch, env, _abi, fixed_actions, N, flop_exec, launches=200, fused_k=20, fused_launches=20):
    """after the timed region, same env (its state simply moves on): the env-only single-launch loop
    and the fused 20-step rollout — the two lines that show the kernel rather than the actor"""
    res = {}
    for name, fn, per in (("single_launch", lambda: env.step(fixed_actions), 1),
                          (f"fused_{fused_k}", lambda: env.rollout(fused_k, _abi.POLICY_RANDOM), fused_k)):
        n = launches if per == 1 else fused_launches
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        env.profile_begin()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        ms, cnt = env.profile_end()
        sec = ms / cnt * 1e-3
        res[name] = {"env_steps_per_s_wall": N * per * n / wall, "kernel_us_per_policy_step": sec / per * 1e6,
                     "valu_frac": flop_exec * N * per / sec / 1e12 / FP32_VECTOR_PEAK_TFLOPS,
                     "hbm_frac_algorithmic": ALGO_BYTES_PER_ENV_STEP * N * per / sec / 1e9 / HBM_PEAK_GBS, "launches": n}
    return res


def trainer_loop(torch, actor, N, variant, device, steps=300):
    """The loop the reference's trainer runs around the env (src/my_trainers/base_KA_DDPG.py:170-262) through the Python mirror of
    its interface: actor -> VecEnv.step, then — after EVERY step — the 26 env-0 diagnostics, wheel_torqs / action_scale and every
    value of infos["episode"] with .item(). Three lines, `steps` steps each, after the timed region: without the reads, with a bare
    stream synchronisation in their place (the floor of ANY per-step host read: the host can no longer run ahead of the GPU), and
    with the reads. tools/trainer_loop_probe.py is the longer form (profiles/*_trainer_loop.txt)."""
    from options_in_simulation_amd import VecEnv
    from options_in_simulation_amd.tracking import ENV0_TAGS
    env = VecEnv(num_envs=N, device=str(device), variant=variant, seed=1)

    def loop(reads, bare_sync):
        states, _ = env.reset()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            with torch.no_grad():
                actions = actor(states)
            states, _rew, _term, _trunc, infos = env.step(actions)
            if bare_sync:
                torch.cuda.current_stream().synchronize()
            if reads:
                _ = [env.task.observed_components[k] for k in ENV0_TAGS]      # :184-209
                _ = env.task.wheel_torqs / env.task.action_scale             # :212
                for _k, v in infos["episode"].items():                       # :259-262
                    if isinstance(v, torch.Tensor) and v.numel() == 1:
                        v.item()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps * 1e6

    loop(True, False)
    res = {"what": "reference trainer's per-step loop around the env (base_KA_DDPG.py:170-262) through VecEnv, us per step",
           "steps_per_line": steps, "no_reads": loop(False, False), "bare_stream_synchronisation": loop(False, True),
           "with_the_trainers_reads": loop(True, False)}
    env.close()
    return res


def merge_raw(sts):
    """the finished-episode statistics of several shards of one rank as one record (sums add, extremes combine)"""
    from options_in_simulation_amd import _abi
    m = _abi.OisStats()
    m.return_min, m.return_max = float("inf"), float("-inf")
    for s in sts:
        m.episodes += s.episodes; m.return_sum += s.return_sum; m.return_sqsum += s.return_sqsum
        m.length_sum += s.length_sum; m.steps += s.steps
        m.return_min, m.return_max = min(m.return_min, s.return_min), max(m.return_max, s.return_max)
    return m


def rank_rows(rows):
    """[[rank, device, arena_offset, arenas, elapsed_s, kernel_ms], ...] (one all-gather) -> the rank table of the line"""
    return [{"rank": int(r), "device": int(d), "arena_offset": int(o), "arenas": int(n), "elapsed_s": float(e), "kernel_ms": float(k)}
            for r, d, o, n, e, k in rows]


def scaling_detail(table, steps, per_launch):
    """per-GPU rates of the same run, so that a SCALE line can be read without a second run: each rank's own env-steps/s over
    ITS elapsed time, and which rank set the (max-over-ranks) time the headline value uses"""
    rates = [t["arenas"] * steps * per_launch / t["elapsed_s"] if t["elapsed_s"] > 0 else None for t in table]
    slow = max(table, key=lambda t: t["elapsed_s"])
    return {"per_gpu_env_steps_per_s": rates, "slowest_rank": slow["rank"], "slowest_elapsed_s": slow["elapsed_s"],
            "fastest_elapsed_s": min(t["elapsed_s"] for t in table)}


def dry_run(args, world, rank):
    """OIS_BENCH_DRYRUN=1: the launch / rendezvous / collective plumbing of a run without a GPU
    (tests/test_bench_launch.py): every rank joins a gloo group, the rank table goes through the
    same all-gather as in a real run, rank 0 prints one JSON line."""
    import torch
    import torch.distributed as dist
    from options_in_simulation_amd.dist import shard_range
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = shard_range(args.arenas * world, rank, world)
    t0 = time.perf_counter()
    time.sleep(0.01 * (rank + 1))   # a stand-in for the timed region: the slowest rank is the last one
    mine = torch.tensor([rank, -1, lo, hi - lo, time.perf_counter() - t0, 0.0], dtype=torch.float64)
    allr = torch.empty(world * 6, dtype=torch.float64)
    if world > 1:
        dist.all_gather_into_tensor(allr, mine)
    else:
        allr.copy_(mine)
    if rank == 0:
        table = rank_rows(allr.view(world, 6).tolist())
        print(json.dumps({"dryrun": True, "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "effective_warmup": args.warmup + args.prewarm,
                          "config": {"ranks": dist.get_world_size() if world > 1 else 1, "rank_table": table},
                          "scaling_detail": scaling_detail(table, args.steps, 1)}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse_args(argv)
    maybe_self_launch(args, argv)

    import torch
    import torch.distributed as dist
    from options_in_simulation_amd import _abi, build_config
    from options_in_simulation_amd.dist import gather_stats, merge_stats
    from options_in_simulation_amd.vecenv import NativeEnv

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} does not match the launch's WORLD_SIZE={world}")
    if os.environ.get("OIS_BENCH_DRYRUN"):
        return dry_run(args, world, rank)
    # one process per GPU. OIS_DIST_BACKEND=gloo + fewer GPUs than ranks is only for rehearsing the
    # multi-process path on a 1-GPU box (ranks then share device 0; RCCL refuses duplicate GPUs).
    backend = os.environ.get("OIS_DIST_BACKEND", "nccl")
    # preflight, before anything initialises a GPU or joins a group: a rank per GPU needs a GPU per rank. Counting devices does
    # not touch them. On failure every rank prints the table it would have run (rank -> device -> arenas) and exits non-zero —
    # an exit, never a re-exec, and before the rendezvous, so that no rank is left waiting in init_process_group
    ndev = torch.cuda.device_count()
    if backend != "gloo" and ndev < world:
        from options_in_simulation_amd.dist import shard_range
        rows = [{"rank": r, "device": r, "present": r < ndev, "arena_offset": shard_range(args.arenas * world, r, world)[0],
                 "arenas": args.arenas} for r in range(world)]
        if rank == 0:
            print(json.dumps({"error": f"--gpus {world} needs {world} visible GPUs, torch.cuda.device_count() = {ndev} "
                                       f"(HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES = {os.environ.get('HIP_VISIBLE_DEVICES') or os.environ.get('ROCR_VISIBLE_DEVICES')}); "
                                       "OIS_DIST_BACKEND=gloo rehearses the ranks on fewer GPUs", "rank_table": rows}), file=sys.stderr, flush=True)
        raise SystemExit(3)
    dev_index = local_rank % max(1, ndev) if backend == "gloo" else local_rank
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    comm_device = torch.device("cpu") if backend == "gloo" else device
    # OIS_BENCH_FORCE_DIST=1: join a process group even with one rank, so that a 1-GPU box runs every
    # collective of the N-rank path through RCCL (rehearsal; the line then still says n_gpus = 1)
    dist_on = world > 1 or bool(os.environ.get("OIS_BENCH_FORCE_DIST"))
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(free_port()))
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    N = args.arenas
    shards = max(1, args.shards)
    if N % shards or ((N // shards) % 16 and shards > 1):
        raise SystemExit("--shards must divide the arena count into multiples of 16 (the reduction groups)")
    if shards > 1 and (args.fused or args.terrain):
        raise SystemExit("--shards is an A/B of the single-step actor loop")
    n_shard = N // shards
    cfgs = [build_config(variant=args.variant, num_envs=n_shard, seed=42, arena_offset=rank * N + i * n_shard) for i in range(shards)]
    cfg = cfgs[0]
    terrain_note = "flat terrain"
    if args.domain_rand:
        terrain_note = "per-arena friction U(0.5,1.0) + plane slope {-0.3,-0.2,0,0.2,0.3}"
    if args.terrain:
        terrain_note = "height-field terrain, 6 x 2 random tiles of 10 m (stairs +-0.15 m, obstacles, pyramid slopes)" + (
            ", footprints sampled every 2nd substep (ois_terrain.footprint_stride = 2)" if args.footprint_stride == 2 else "")

    def make_env(c=None):
        c = cfg if c is None else c
        e = NativeEnv(c, device)
        if args.domain_rand:
            e.randomize_params(0.5, 1.0, [-0.3, -0.2, 0.0, 0.2, 0.3])  # origin_terrain_generator.py:92
        if args.terrain:
            import numpy as np
            from options_in_simulation_amd import terrain as tg
            T = tg.Terrain({"numLevels": 6, "numTerrains": 2, "mapLength": 10.0, "mapWidth": 10.0}, N, seed=42)
            e.attach_terrain(T, T.sample_levels_types(N, rng=np.random.RandomState(rank)), footprint_stride=args.footprint_stride)
        e.reset(42)
        return e

    actor = None if args.no_actor else build_actor(torch, device)
    K_fused = max(0, args.fused)
    if K_fused:
        actor = None

    def loop(envs, obs_list, fixed_list, streams, steps):
        """the benchmark's step, `steps` times: actor (or fixed actions) -> ois_step (or a fused rollout) per shard"""
        for _ in range(steps):
            for e, o, f, st in zip(envs, obs_list, fixed_list, streams):
                ctx = torch.cuda.stream(st) if st is not None else contextlib.nullcontext()
                with ctx, torch.no_grad():
                    if K_fused:
                        e.rollout(K_fused, _abi.POLICY_RANDOM)
                    else:
                        e.step(f if actor is None else actor(o))

    def build_shards():
        envs = [make_env(c) for c in cfgs]
        obs_l = [e.buffer(_abi.BUF_OBS) for e in envs]
        fixed = [torch.rand((n_shard, cfg.action_dim), device=device, generator=torch.Generator(device).manual_seed(1 + i))
                 for i in range(shards)]
        streams = [torch.cuda.Stream(device) for _ in range(shards)] if shards > 1 else [None]
        # reset, parameter draws and the fixed actions above were enqueued on the current stream; the shards' own streams do
        # not order against it
        for st in streams:
            if st is not None:
                st.wait_stream(torch.cuda.current_stream(device))
        return envs, obs_l, fixed, streams

    def shard_stats(e, st, reset_after=True):
        """ois_stats_fetch of a shard ON the stream its steps run on (the fetch synchronises that stream)"""
        if st is None:
            return e.stats(reset_after=reset_after)
        with torch.cuda.stream(st):
            return e.stats(reset_after=reset_after)

    # What a cold process gives, and the steady state before the contract's W warm-up steps. A SCRATCH set of environments of
    # the same configuration first does exactly the contract's W + K steps, timed (`cold_start`: clocks, hipBLASLt's kernel
    # choice and the code caches are still settling), then `--prewarm` more untimed iterations. The measured environments'
    # state (gamma schedule, episode phase) is exactly what W + K steps make it.
    cold = None
    if args.prewarm > 0:
        sc = build_shards()
        loop(*sc, args.warmup)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        loop(*sc, args.steps)
        torch.cuda.synchronize()
        cold = (time.perf_counter() - t0) / args.steps * 1e3
        loop(*sc, args.prewarm)
        torch.cuda.synchronize()
        for e in sc[0]:
            e.close()
        del sc
    envs, obs_l, fixed_l, streams = build_shards()
    env, fixed_actions = envs[0], fixed_l[0]

    def barrier():
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()

    loop(envs, obs_l, fixed_l, streams, args.warmup)
    if dist_on:  # warm the collective too
        gather_stats(shard_stats(env, streams[0]), device)
        for e, st in zip(envs[1:], streams[1:]):
            shard_stats(e, st)
    barrier()
    for e in envs:
        e.profile_begin()  # start / stop events on every step kernel's own dispatch packet
    t0 = time.perf_counter()
    gathered = None
    for s in range(args.steps):
        loop(envs, obs_l, fixed_l, streams, 1)
        if (s + 1) % args.log_interval == 0 or s + 1 == args.steps:
            sts = [shard_stats(e, st) for e, st in zip(envs, streams)]      # one tiny D2H per logging interval (and shard)
            st = sts[0] if shards == 1 else merge_raw(sts)
            gathered = gather_stats(st, device) if dist_on else [st]
    barrier()
    elapsed = time.perf_counter() - t0
    my_elapsed = elapsed
    prof = [e.profile_end() for e in envs]
    kernel_ms_total, launches = sum(p[0] for p in prof), sum(p[1] for p in prof)
    kernel_ms = kernel_ms_total / max(1, launches)
    # the same K steps without the event pairs (hipLaunchKernel instead of hipExtLaunchKernel + two events): what the
    # instrumentation of the timed region costs
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    loop(envs, obs_l, fixed_l, streams, args.steps)
    torch.cuda.synchronize()
    uninstrumented_ms = (time.perf_counter() - t1) / args.steps * 1e3
    if dist_on:
        t = torch.tensor([elapsed], dtype=torch.float64, device=comm_device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        # who took part and how long each took: (rank, device, arena_offset, arenas, elapsed_s, kernel_ms) through one all-gather
        mine = torch.tensor([rank, dev_index, cfg.arena_offset, N, my_elapsed, kernel_ms], dtype=torch.float64, device=comm_device)
        allr = torch.empty(world * 6, dtype=torch.float64, device=comm_device)
        dist.all_gather_into_tensor(allr, mine)
        rank_table = rank_rows(allr.view(world, 6).cpu().tolist())
    else:
        rank_table = rank_rows([[0, dev_index, int(cfg.arena_offset), N, my_elapsed, kernel_ms]])

    per_launch = max(1, K_fused)
    total_env_steps = float(N) * world * args.steps * per_launch
    value = total_env_steps / elapsed

    if rank == 0:
        kinfo = env.kernel_info()
        kid = kinfo.split(" ")[0]
        nsub = cfg.decimation + cfg.extra_substeps
        flop_exec, valu_exec, flop_prov = executed_flop_per_env_step(kid, nsub)
        flop_algo = algorithmic_flop_per_env_step()
        arenas_per_launch = n_shard
        flops = flop_exec * arenas_per_launch * per_launch / (kernel_ms * 1e-3) / 1e12
        flops_algo = flop_algo * arenas_per_launch * per_launch / (kernel_ms * 1e-3) / 1e12
        hbm = ALGO_BYTES_PER_ENV_STEP * arenas_per_launch * per_launch / (kernel_ms * 1e-3) / 1e9
        copy_gbs = measured_copy_gbs(torch, device)  # after the timed region
        plain = not (args.terrain or K_fused or shards > 1)
        traffic, traffic_prov = pmc_traffic(N) if plain else (None, {"status": "not quoted for this workload"})
        sq, sq_prov = pmc_sq(N) if (plain and actor is not None) else (None, {"status": "not quoted for this workload"})
        gamma_note = ""
        if args.variant in ("KAMMA", "KAMMA_medium", "KAMMA_curr1", "KAMMA_curr2") and actor is not None:
            g = 1.0 - (args.warmup + args.steps + 1) / float(cfg.max_sim_steps)
            gamma_note = (f"; gamma1 >= {g:.4f} over the timed window, so the switch executes the expert's torques on nearly "
                          "every step and the actor's output is computed but rarely applied (same cost either way)")
        out = {
            "metric": "env-steps/sec at 65 536 parallel arenas; achieved HBM GB/s vs roofline",
            "value": value, "unit": "env-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "effective_warmup": args.warmup + args.prewarm,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{N} arenas/GPU, {terrain_note}, {args.variant} task, "
                                   + (f"in-kernel random-torque policy, {K_fused} policy steps per launch (env only)" if K_fused else
                                      "fixed random actions (env only)" if actor is None else
                                      "scripted actor MLP 6-512-512-128-4 (fp32 PyTorch-ROCm) inference in-loop")
                                   + (f", as {shards} shards of {n_shard} arenas on {shards} HIP streams" if shards > 1 else "")
                                   + (" [BASELINE.json configs[3]]" if (args.domain_rand or args.terrain) else " [BASELINE.json configs[2]]")
                                   + gamma_note,
                       "arenas_per_gpu": N, "shards": shards, "prewarm_scratch_iterations": args.prewarm, "substeps_per_step": nsub,
                       "solver_iterations": cfg.solver_iterations, "sharding": f"arenas x{world}, no data-path collective",
                       "stats_allgather_every": args.log_interval,
                       "ranks": dist.get_world_size() if dist_on else 1,
                       "backend": (dist.get_backend() + (" (RCCL)" if backend == "nccl" else "")) if dist_on else "none",
                       "rank_table": rank_table},
            "warmup_detail": {"contract_warmup_steps": args.warmup, "prewarm_iterations_on_scratch_environments": args.prewarm,
                              "cold_start_ms_per_step": cold,
                              "cold_start_what": "the same W + K steps in this process BEFORE anything was warm (scratch environments, then "
                                                 "the pre-warm iterations): what a run with only the contract's W warm-up steps would report",
                              "uninstrumented_ms_per_step": uninstrumented_ms,
                              "uninstrumented_what": "K more steps right after the timed region without the two events per launch "
                                                     "(plain hipLaunchKernel): the cost of timing the kernel inside the timed region"},
            "roofline": {"bound": "valu", "achieved": flops, "peak": FP32_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": flops / FP32_VECTOR_PEAK_TFLOPS, "traffic": traffic,
                         "kernel": kinfo, "kernel_ms": kernel_ms, "kernel_launches": launches, "arenas_per_launch": arenas_per_launch,
                         "kernel_ms_how": "mean over the timed region of hipEventElapsedTime(start, stop) with both events "
                                          "attached to the kernel's dispatch packet (hipExtLaunchKernel): kernel begin -> end as "
                                          "rocprofv3 --kernel-trace measures it, no launch gap",
                         "flop_per_env_step": flop_exec, "flop_provenance": flop_prov,
                         "flop_how": "EXECUTED fp32 operations of the launched instantiation: ISA instruction mix of its substep loop x "
                                     "substeps (tools/isa_census.py, fma = 2, packed = x 2); a kernel that issues more instructions scores "
                                     "higher on this figure, hence the two instruction-independent ones beside it",
                         "flop_algorithmic_per_env_step": flop_algo,
                         "flop_algorithmic_how": "operations of the MODEL counted from docs/SPEC.md section 3.11 (tools/spec_flops.py): "
                                                 "20 substeps x 1 653 + 171 for the policy step's own arithmetic",
                         "algorithmic": {"achieved": flops_algo, "frac": flops_algo / FP32_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s"},
                         "executed_over_algorithmic": flop_exec / flop_algo,
                         "valu_issue": sq, "valu_issue_provenance": sq_prov,
                         "valu_issue_live": None if valu_exec is None else {
                             "valu_instructions_per_lane_per_env_step_substeps_only": valu_exec,
                             "issue_limited_us_at_2_GHz": valu_exec * 4.0 / 2.0e3,
                             "issue_slot_utilisation_at_2_GHz": valu_exec * 4.0 / 2.0e3 / (kernel_ms * 1e3),
                             "what": "substep-loop VALU instructions of the census x 4 cycles (one wave per SIMD issues one VALU instruction per "
                                     "4 cycles) / 2.0 GHz over this run's event-timed kernel duration; `valu_issue` has the counters"},
                         "hbm": {"achieved": hbm, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": hbm / HBM_PEAK_GBS,
                                 "algorithmic_bytes_per_launch": ALGO_BYTES_PER_ENV_STEP * arenas_per_launch * per_launch,
                                 "measured_copy": {"GB/s": copy_gbs, "frac": hbm / copy_gbs,
                                                   "what": "1 GiB device-to-device torch copy on this GPU, read + write bytes / time"}},
                         "traffic_provenance": traffic_prov,
                         "note": "fp32-VALU issue bound, not HBM bound: 20 substeps x 4-iteration contact solve = ~12k VALU "
                                 "instructions per env-step, more than half of them packed; one wave per SIMD at 65 536 arenas "
                                 "(1024 waves on 1024 SIMDs). BASELINE.md's >= 50 % HBM target would need <= 140 lane-instructions "
                                 "per arena-substep; DESIGN.md section 4"},
            "env_only_env_steps_per_s": arenas_per_launch * per_launch / (kernel_ms * 1e-3),
            "kernel_src_sha256": kernel_source_hash(),
            "scaling_detail": scaling_detail(rank_table, args.steps, per_launch),
        }
        if gathered:
            m = merge_stats(gathered)
            out["episode_stats"] = {"episodes": m["episodes"], "return_mean": m["return_mean"], "return_std": m["return_std"],
                                    "episodes_per_rank": [g.episodes for g in gathered]}
        if not args.no_extras and not K_fused and not args.terrain and shards == 1:
            out["env_only"] = side_measurements(torch, env, _abi, fixed_actions, N, flop_exec)
            if world == 1 and not args.no_actor:
                out["trainer_loop"] = trainer_loop(torch, actor, N, args.variant, device)
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(N)
            except Exception as exc:  # noqa: BLE001 — the GPU line must survive a broken checker build
                out["cpu_baseline"] = {"value": None, "unit": "env-steps/s", "cores": 0, "kind": "port",
                                       "sample": f"not measured: {type(exc).__name__}: {exc}"}
        print(json.dumps(out), flush=True)
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()
    for e in envs:
        e.close()


if __name__ == "__main__":
    main()
