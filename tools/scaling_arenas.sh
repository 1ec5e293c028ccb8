#!/bin/bash
# bench.py --no-actor (single ois_step launches) and --fused 20 at several arena counts -> gpurun_out/<R>_scaling_arenas.json
set -e -o pipefail
R=${1:-r1}
OUT=$PWD/gpurun_out
for N in 4096 16384 32768 65536 131072 262144 524288; do
  python bench.py --arenas $N --no-actor --no-cpu-baseline --no-extras --steps 300 --warmup 30 > $OUT/b_$N.json 2>/dev/null
  python bench.py --arenas $N --fused 20 --no-cpu-baseline --steps 30 --warmup 5 > $OUT/f_$N.json 2>/dev/null
  echo "N=$N done"
done
python3 - "$OUT" "$R" <<'PY'
import json, sys
OUT, R = sys.argv[1:3]
rows = []
for N in (4096, 16384, 32768, 65536, 131072, 262144, 524288):
    d = json.loads(open(f"{OUT}/b_{N}.json").readline())
    f = json.loads(open(f"{OUT}/f_{N}.json").readline())
    rows.append({"arenas": N, "kernel": d["roofline"]["kernel"].split(" ")[0], "env_steps_per_s": d["value"], "kernel_us": 1e3 * d["roofline"]["kernel_ms"],
                 "valu_frac_executed": d["roofline"]["frac"], "valu_frac_algorithmic": d["roofline"]["algorithmic"]["frac"], "hbm_frac": d["roofline"]["hbm"]["frac"],
                 "fused20_env_steps_per_s": f["value"], "fused20_kernel_us_per_policy_step": 1e3 * f["roofline"]["kernel_ms"] / 20,
                 "fused20_valu_frac_executed": f["roofline"]["frac"], "kernel_src_sha256": d["kernel_src_sha256"]})
json.dump({"what": "bench.py --no-actor (single ois_step launches, fixed random actions) and --fused 20 (in-kernel random policy) at several arena counts, one MI355X",
           "rows": rows}, open(f"{OUT}/{R}_scaling_arenas.json", "w"), indent=1)
for r in rows: print(r)
PY
