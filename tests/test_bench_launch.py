"""bench.py's own rank launch: `python bench.py --gpus N` must work without a
torch.distributed environment — it starts its N ranks itself, before touching the GPU — and the
JSON line must show that N ranks took part. No GPU here: OIS_BENCH_DRYRUN keeps the launch,
rendezvous and all-gather and skips the environment."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _clean_env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(extra)
    return env


def test_spawn_command_is_the_drivers_launch_line():
    import bench
    cmd = bench.spawn_command(4, ["--gpus", "4", "--steps", "7"], 29511)
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"]
    assert "--nproc-per-node=4" in cmd and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[cmd.index("--master-port") + 1] == "29511"
    assert cmd[-5] == os.path.join(ROOT, "bench.py") and cmd[-4:] == ["--gpus", "4", "--steps", "7"]


def test_no_self_launch_for_one_gpu_or_inside_a_launch(monkeypatch):
    import bench
    called = []
    monkeypatch.setattr(bench.subprocess, "call", lambda *a, **k: called.append(a) or 0)
    monkeypatch.delenv("RANK", raising=False)
    monkeypatch.delenv("LOCAL_RANK", raising=False)
    bench.maybe_self_launch(bench.parse_args(["--gpus", "1"]), ["--gpus", "1"])
    monkeypatch.setenv("RANK", "0")
    bench.maybe_self_launch(bench.parse_args(["--gpus", "2"]), ["--gpus", "2"])  # already a rank of somebody's launch
    assert not called


def test_bare_bench_gpus_2_starts_two_ranks_and_prints_one_line():
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                          "--arenas", "1000"], env=_clean_env(OIS_BENCH_DRYRUN="1"), capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["ranks"] == 2 and d["steps"] == 3
    table = d["config"]["rank_table"]
    assert [r["rank"] for r in table] == [0, 1]
    assert [(r["arena_offset"], r["arenas"]) for r in table] == [(0, 1000), (1000, 1000)]


def test_bare_bench_gpus_8_starts_eight_ranks_with_a_timed_rank_table():
    """the driver's 8-GPU launch shape on this container's 8 cores (gloo, no GPU): one line, ranks: 8, eight rows with each
    rank's own elapsed time and kernel time through the one all-gather, and the per-GPU rates / slowest rank that let a SCALE
    line be read without a second run"""
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "3", "--warmup", "1",
                          "--arenas", "65536"], env=_clean_env(OIS_BENCH_DRYRUN="1", OMP_NUM_THREADS="1"), capture_output=True, text=True,
                         timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["config"]["ranks"] == 8
    table = d["config"]["rank_table"]
    assert [r["rank"] for r in table] == list(range(8))
    assert [(r["arena_offset"], r["arenas"]) for r in table] == [(i * 65536, 65536) for i in range(8)]   # BASELINE configs[4]: 524 288
    assert all(r["elapsed_s"] > 0 and "kernel_ms" in r for r in table)
    sd = d["scaling_detail"]
    assert len(sd["per_gpu_env_steps_per_s"]) == 8 and sd["slowest_rank"] == 7   # the dry run sleeps 10 ms x (rank + 1)
    assert sd["slowest_elapsed_s"] >= sd["fastest_elapsed_s"] > 0
    assert d["effective_warmup"] == d["warmup"] + 40


def test_world_size_mismatch_is_an_error_not_a_silent_run():
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"],
                         env=_clean_env(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", OIS_BENCH_DRYRUN="1"),
                         capture_output=True, text=True, timeout=300)
    assert res.returncode != 0 and "WORLD_SIZE" in (res.stderr + res.stdout)


def test_fewer_gpus_than_ranks_fails_before_the_rendezvous():
    """`--gpus N` with RCCL on a box with fewer than N visible GPUs: every rank exits non-zero BEFORE it joins the process group
    (a lone rank 0 started here would otherwise wait for its peers until the rendezvous times out) and rank 0 says which rank
    would have used which device and arenas. Here: no GPU at all, world size 2, only rank 0 started."""
    import time
    t0 = time.time()
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1"],
                         env=_clean_env(RANK="0", LOCAL_RANK="0", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT="29533"),
                         capture_output=True, text=True, timeout=300)
    assert res.returncode == 3 and time.time() - t0 < 120
    err = json.loads([l for l in res.stderr.splitlines() if l.startswith("{")][-1])
    assert "needs 2 visible GPUs" in err["error"] and "device_count() = 0" in err["error"]
    assert [(r["rank"], r["device"], r["present"], r["arena_offset"], r["arenas"]) for r in err["rank_table"]] == \
        [(0, 0, False, 0, 65536), (1, 1, False, 65536, 65536)]
    assert not [l for l in res.stdout.splitlines() if l.startswith("{")]
