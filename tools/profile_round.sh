#!/bin/bash
# Regenerates the measured artefacts of one round on the GPU box (run through gpurun from the
# repo root): bench lines, the rocprofv3 kernel-trace summary of the same command, and the HBM
# traffic counters (separate --pmc passes, as the microarch guide prescribes).
#   git rev-parse HEAD > .git_head && gpurun --timeout 1200 -- 'bash tools/profile_round.sh r4'
#        -> gpurun_out/r4_*   (copy what is judged into profiles/). .git_head travels with the snapshot (the box has no .git);
#        every artefact also names the sha256 of the kernel sources it was taken on, which is what bench.py checks.
set -e -o pipefail
R=${1:-r4}
OUT=$PWD/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
P=/tmp/ois_prof_$$
rm -rf $P && mkdir -p $P
HASH=$(python3 -c "import bench; print(bench.kernel_source_hash())")
echo "kernel sources sha256 $HASH" > $OUT/${R}_kernel_src_sha256.txt
# the issue counters of single launches in the actor loop: bench.py quotes `roofline.valu_issue` from profiles/${R}_pmc_sq.json
bash tools/profile_sq.sh $R > $OUT/${R}_sq.log 2>&1 || { tail -5 $OUT/${R}_sq.log; exit 1; }
cp $OUT/${R}_pmc_sq.json profiles/${R}_pmc_sq.json
bash tools/profile_sq.sh $R fused >> $OUT/${R}_sq.log 2>&1 || { tail -5 $OUT/${R}_sq.log; exit 1; }
echo "sq counters done"
# the traffic counters: bench.py quotes `traffic` from profiles/${R}_pmc_traffic.json when its source hash is this checkout's
for C in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && rocprofv3 --pmc $C --kernel-trace --output-format csv -d $P/$C -- python3 $OLDPWD/bench.py --steps 60 --warmup 10 --no-actor --no-cpu-baseline --no-extras > /dev/null 2> $P/$C.err) || { tail -5 $P/$C.err; exit 1; }
  cp $(find $P/$C -name '*counter_collection.csv' | head -1) $P/$C.csv
done
python3 - "$P" "$OUT" "$R" "$PWD" <<'PY'
import csv, json, subprocess, sys
P, OUT, R, REPO = sys.argv[1:5]
sys.path.insert(0, REPO)
import bench
head = open(REPO + "/.git_head").read().strip() if __import__("os").path.exists(REPO + "/.git_head") else None
res = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    vals = []
    rows = list(csv.DictReader(open(f"{P}/{c}.csv")))
    for r in rows:
        if "step_kernel" in r.get("Kernel_Name", "") and r.get("Counter_Name") == c:
            vals.append(float(r["Counter_Value"]))
    res[c] = vals
    with open(f"{OUT}/{R}_pmc_{c.split('_')[0].lower()}_sample.csv", "w") as f:
        w = csv.writer(f); w.writerow(["Kernel_Name", "Counter_Name", "Counter_Value"])
        for r in [x for x in rows if "step_kernel" in x.get("Kernel_Name", "")][:12]:
            w.writerow([r["Kernel_Name"][:60], r["Counter_Name"], r["Counter_Value"]])
f, w = res["FETCH_SIZE"], res["WRITE_SIZE"]
fm, wm = sum(f) / len(f), sum(w) / len(w)
out = {"FETCH_SIZE_KB_per_launch_mean": fm, "FETCH_SIZE_launches": len(f), "WRITE_SIZE_KB_per_launch_mean": wm,
       "WRITE_SIZE_launches": len(w), "fetch_correction": 2.0,
       "hbm_bytes_per_launch": (2.0 * fm + wm) * 1024.0, "arenas": 65536,
       "bytes_per_arena_step": (2.0 * fm + wm) * 1024.0 / 65536, "ratio_to_algorithmic_352": (2.0 * fm + wm) * 1024.0 / 65536 / 352.0,
       "kernel_src_sha256": bench.kernel_source_hash(), "git_head": head,
       "kernel": "step_kernel<1,1,4,1> (one arena per lane, packed fp32 over wheel pairs, flat plane, 1 policy step per launch)",
       "command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace --output-format csv -- python3 bench.py --steps 60 "
                  "--warmup 10 --no-actor --no-cpu-baseline (two separate passes; FETCH_SIZE x2 = the guide's gfx950 correction)"}
json.dump(out, open(f"{OUT}/{R}_pmc_traffic.json", "w"), indent=1)
print(json.dumps(out))
PY
cp $OUT/${R}_pmc_traffic.json profiles/${R}_pmc_traffic.json
python bench.py > $OUT/${R}_bench.json 2> $OUT/${R}_bench.err
echo "bench done" && tail -c 300 $OUT/${R}_bench.json
python bench.py --no-actor --no-cpu-baseline --steps 400 --warmup 40 > $OUT/${R}_bench_envonly.json 2>> $OUT/${R}_bench.err
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $P/kt -- python3 $OLDPWD/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-extras > $OUT/${R}_bench_under_rocprof.json 2> $P/kt.err) || { tail -5 $P/kt.err; exit 1; }
cp $(find $P/kt -name '*kernel_stats.csv' | head -1) $OUT/${R}_bench_kernel_stats.csv
echo "kernel trace done"
grep step_kernel $OUT/${R}_bench_kernel_stats.csv | cut -c1-200
# the multi-process path on this 1-GPU box: two gloo ranks sharing the card, launched by bench.py itself
OIS_DIST_BACKEND=gloo python bench.py --gpus 2 --steps 120 --warmup 10 > $OUT/${R}_bench_gloo2.json 2>> $OUT/${R}_bench.err
tail -c 400 $OUT/${R}_bench_gloo2.json
# side measurements of the round (each file starts with the kernel-source hash it was taken on)
(echo "# kernel sources sha256 $HASH"; python tools/launch_overhead_probe.py 4096 16384 65536) > $OUT/${R}_launch_overhead.txt 2>&1
(echo "# kernel sources sha256 $HASH"; python tools/inloop_probe.py) > $OUT/${R}_inloop_probe.txt 2>&1
(echo "# kernel sources sha256 $HASH"; python tools/level_cost_probe.py) > $OUT/${R}_level_cost.txt 2>&1
python bench.py --fused 20 --steps 60 --warmup 5 --no-cpu-baseline > $OUT/${R}_bench_fused20.json 2>> $OUT/${R}_bench.err
python bench.py --domain-rand --no-cpu-baseline > $OUT/${R}_bench_config4_domain_rand.json 2>> $OUT/${R}_bench.err
python bench.py --terrain --no-cpu-baseline > $OUT/${R}_bench_config4_terrain.json 2>> $OUT/${R}_bench.err
python bench.py --terrain --footprint-stride 2 --no-cpu-baseline > $OUT/${R}_bench_config4_terrain_stride2.json 2>> $OUT/${R}_bench.err
python tools/terrain_probe.py > /dev/null 2>&1 && cp $OUT/terrain_probe.json $OUT/${R}_terrain_probe.json
python bench.py --steps 20 --warmup 5 > $OUT/${R}_bench_driver_shape.json 2>> $OUT/${R}_bench.err
# the reference trainer's per-step loop around the env (base_KA_DDPG.py:170-262): the four lines + where the host spends a step,
# and the rocprofv3 kernel trace of the same loop
(echo "# kernel sources sha256 $HASH"; python tools/trainer_loop_probe.py 65536 --breakdown; python tools/trainer_loop_probe.py 65536) 2>&1 | grep -v amdgpu.ids > $OUT/${R}_trainer_loop.txt  # two runs: the host's wake-up moves by +-10 us
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $P/tl -- python3 $OLDPWD/tools/trainer_loop_probe.py > $P/tl.log 2> $P/tl.err) || { tail -5 $P/tl.err; exit 1; }
cp $(find $P/tl -name '*kernel_stats.csv' | head -1) $OUT/${R}_trainer_loop_kernel_stats.csv
(echo "# kernel sources sha256 $HASH"; python tools/info_probe.py) > $OUT/${R}_info_probe.txt 2>&1
python tools/gae_probe.py > /dev/null 2>&1 && cp $OUT/gae_probe.json $OUT/${R}_gae_probe.json
# one wave per SIMD, claimed (OIS_STEP_WAVES=1, the default at this size) against two allowed (=2): the same body
(echo "# kernel sources sha256 $HASH"; bash tools/waves_ab.sh) > $OUT/${R}_ab_occupancy.txt 2>&1
# the N-rank path's collective through RCCL itself, with the one rank this box has
(echo "# kernel sources sha256 $HASH"; python tools/nccl_single_rank_check.py) > $OUT/${R}_rccl_1rank.txt 2>&1
OIS_BENCH_FORCE_DIST=1 python bench.py --steps 120 --warmup 10 --no-cpu-baseline --no-extras > $OUT/${R}_bench_rccl_1rank.json 2>> $OUT/${R}_bench.err
echo "side measurements done"
