#!/bin/bash
# SQ issue counters of the fused rollout (20 policy steps per dispatch): one rocprofv3 --pmc pass
# per counter (never combined with other trace domains), summary -> gpurun_out/<round>_pmc_sq.json
set -e -o pipefail
R=${1:-r1}
OUT=$PWD/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
P=/tmp/ois_sq_$$
rm -rf $P && mkdir -p $P
for C in SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVES GRBM_GUI_ACTIVE; do
  (cd /tmp && rocprofv3 --pmc $C --kernel-trace --output-format csv -d $P/$C -- python3 $OLDPWD/tools/rollout_probe.py --launches 3 --k 20 > /dev/null 2> $P/$C.err) || { tail -5 $P/$C.err; exit 1; }
  cp $(find $P/$C -name '*counter_collection.csv' | head -1) $P/$C.csv
  cp $(find $P/$C -name '*kernel_trace.csv' | head -1) $P/$C.trace.csv
done
python3 - "$P" "$OUT" "$R" <<'PY'
import csv, json, sys
P, OUT, R = sys.argv[1:4]
names = ["SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_WAVES", "GRBM_GUI_ACTIVE"]
res = {}
for c in names:
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(f"{P}/{c}.csv"))
            if "step_kernel" in r.get("Kernel_Name", "") and r.get("Counter_Name") == c]
    res[c] = sum(vals[1:]) / max(1, len(vals[1:]))  # skip the warm-up dispatch
dur = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(f"{P}/SQ_WAVES.trace.csv"))
       if "step_kernel" in r.get("Kernel_Name", "")]
res["dispatch_ns"] = sum(dur[1:]) / max(1, len(dur[1:]))
res["policy_steps_per_dispatch"] = 20
res["valu_instr_per_wave_per_step"] = res["SQ_INSTS_VALU"] / max(1.0, res["SQ_WAVES"]) / 20
res["quad_cycles_per_valu_instr"] = res["SQ_ACTIVE_INST_VALU"] / res["SQ_INSTS_VALU"]
res["valu_active_fraction_of_wave_life"] = res["SQ_ACTIVE_INST_VALU"] / res["SQ_WAVE_CYCLES"]
res["clock_GHz_est"] = 4.0 * res["SQ_WAVE_CYCLES"] / max(1.0, res["SQ_WAVES"]) / res["dispatch_ns"]
res["note"] = ("SQ_* cycle counters are in quad-cycles (4 shader cycles); fused rollout of 20 policy steps per dispatch, "
               "65536 arenas, step_kernel<1,true,4>; one --pmc pass per counter")
json.dump(res, open(f"{OUT}/{R}_pmc_sq.json", "w"), indent=1)
print(json.dumps(res))
PY
