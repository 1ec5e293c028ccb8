#!/bin/bash
# SQ issue counters of the step kernel: one rocprofv3 --pmc pass per counter (never combined with other
# trace domains), summary with the kernel-source hash -> gpurun_out/<round>_pmc_sq[_fused].json
#   tools/profile_sq.sh r3           single ois_step launches inside bench.py's actor loop (the headline's launches)
#   tools/profile_sq.sh r3 fused     the fused rollout (20 policy steps per dispatch)
#   tools/profile_sq.sh r4 terrain | terrain2     the fused rollout on the height field, footprint stride 1 | 2
set -e -o pipefail
R=${1:-r1}
MODE=${2:-actor}
OUT=$PWD/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
P=/tmp/ois_sq_$$
rm -rf $P && mkdir -p $P
for C in SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVES GRBM_GUI_ACTIVE; do
  if [ "$MODE" != actor ]; then
    TER=0; [ "$MODE" = terrain ] && TER=1; [ "$MODE" = terrain2 ] && TER=2
    (cd /tmp && rocprofv3 --pmc $C --kernel-trace --output-format csv -d $P/$C -- python3 $OLDPWD/tools/rollout_probe.py --launches 3 --k 20 --terrain $TER > /dev/null 2> $P/$C.err) || { tail -5 $P/$C.err; exit 1; }
  else
    (cd /tmp && rocprofv3 --pmc $C --kernel-trace --output-format csv -d $P/$C -- python3 $OLDPWD/bench.py --steps 40 --warmup 5 --prewarm 20 --no-cpu-baseline --no-extras > /dev/null 2> $P/$C.err) || { tail -5 $P/$C.err; exit 1; }
  fi
  cp $(find $P/$C -name '*counter_collection.csv' | head -1) $P/$C.csv
  cp $(find $P/$C -name '*kernel_trace.csv' | head -1) $P/$C.trace.csv
done
python3 - "$P" "$OUT" "$R" "$MODE" "$PWD" <<'PY'
import csv, json, sys
P, OUT, R, MODE, REPO = sys.argv[1:6]
sys.path.insert(0, REPO)
import bench
K = 1 if MODE == "actor" else 20
SKIP = 25 if MODE == "actor" else 1   # the warm-up dispatch / the pre-warm launches on the scratch environment
names = ["SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_WAVES", "GRBM_GUI_ACTIVE"]
res = {}
for c in names:
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(f"{P}/{c}.csv"))
            if "step_kernel" in r.get("Kernel_Name", "") and r.get("Counter_Name") == c]
    res[c] = sum(vals[SKIP:]) / max(1, len(vals[SKIP:]))
    res.setdefault("dispatches_averaged", len(vals[SKIP:]))
dur = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(f"{P}/SQ_WAVES.trace.csv"))
       if "step_kernel" in r.get("Kernel_Name", "")]
res["dispatch_ns"] = sum(dur[SKIP:]) / max(1, len(dur[SKIP:]))
res["policy_steps_per_dispatch"] = K
res["valu_instr_per_wave_per_step"] = res["SQ_INSTS_VALU"] / max(1.0, res["SQ_WAVES"]) / K
res["quad_cycles_per_valu_instr"] = res["SQ_ACTIVE_INST_VALU"] / res["SQ_INSTS_VALU"]
res["valu_active_fraction_of_wave_life"] = res["SQ_ACTIVE_INST_VALU"] / res["SQ_WAVE_CYCLES"]
res["clock_GHz_est"] = 4.0 * res["SQ_WAVE_CYCLES"] / max(1.0, res["SQ_WAVES"]) / res["dispatch_ns"]
res["issue_limited_ns"] = res["valu_instr_per_wave_per_step"] * K * 4.0 / res["clock_GHz_est"]
res["issue_slot_utilisation"] = res["issue_limited_ns"] / res["dispatch_ns"]
res["mode"] = ("fused rollout, 20 policy steps per dispatch (tools/rollout_probe.py)" + {"terrain": " on the height field", "terrain2": " on the height field, footprint stride 2"}.get(MODE, "") if MODE != "actor" else
               "single ois_step launches inside bench.py's actor loop (bench.py --steps 40 --warmup 5 --prewarm 20)")
res["arenas"] = 65536
res["kernel_src_sha256"] = bench.kernel_source_hash()
res["note"] = ("SQ_* cycle counters are in quad-cycles (4 shader cycles); 65536 arenas, every dispatch whose name contains step_kernel; one --pmc pass per "
               "counter; issue_limited_ns = instructions per wave x 4 cycles / clock (one wave per SIMD issues one VALU "
               "instruction per 4 cycles), issue_slot_utilisation = that / the dispatch's duration")
json.dump(res, open(f"{OUT}/{R}_pmc_sq{'' if MODE == 'actor' else '_' + MODE}.json", "w"), indent=1)
print(json.dumps(res))
PY
