#!/bin/bash
# bench.py --no-actor (single ois_step launches) at several arena counts -> gpurun_out/<R>_scaling_arenas.json
set -e -o pipefail
R=${1:-r1}
OUT=$PWD/gpurun_out
for N in 4096 16384 32768 65536 131072 262144 524288; do
  python bench.py --arenas $N --no-actor --no-cpu-baseline --steps 300 --warmup 30 > $OUT/b_$N.json 2>/dev/null
done
python3 - "$OUT" "$R" <<'PY'
import json, sys
OUT, R = sys.argv[1:3]
rows = []
for N in (4096, 16384, 32768, 65536, 131072, 262144, 524288):
    d = json.loads(open(f"{OUT}/b_{N}.json").readline())
    rows.append({"arenas": N, "env_steps_per_s": d["value"], "kernel_us": 1e3 * d["roofline"]["kernel_ms"],
                 "hbm_frac": d["roofline"]["frac"], "valu_frac": d["roofline"]["valu"]["frac"]})
json.dump({"what": "bench.py --no-actor (single ois_step launches, fixed random actions) at several arena counts, one MI355X",
           "rows": rows}, open(f"{OUT}/{R}_scaling_arenas.json", "w"), indent=1)
for r in rows: print(r)
PY
