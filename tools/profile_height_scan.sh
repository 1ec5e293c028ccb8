#!/bin/bash
# L1 <-> L2 line traffic of ois_height_scan (one rocprofv3 --pmc pass per counter): requests per scan point against the
# floor the tile format allows -> gpurun_out/<round>_height_scan_pmc.json
set -e -o pipefail
R=${1:-r3}
OUT=$PWD/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
P=/tmp/ois_hs_$$
rm -rf $P && mkdir -p $P
(cd /tmp && rocprofv3 --list-avail > $P/avail.txt 2>&1) || true
grep -o "TCP_[A-Z_0-9]*\|TCC_[A-Z_0-9]*" $P/avail.txt | sort -u > $OUT/${R}_counters_tcp_tcc.txt || true
for C in TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum; do
  if (cd /tmp && rocprofv3 --pmc $C --kernel-trace --output-format csv -d $P/$C -- python3 $OLDPWD/tools/height_scan_probe.py 65536 10 > /dev/null 2> $P/$C.err); then
    cp $(find $P/$C -name '*counter_collection.csv' | head -1) $P/$C.csv
  else
    echo "$C: not collected"; tail -2 $P/$C.err
  fi
done
python3 - "$P" "$OUT" "$R" "$PWD" <<'PY'
import csv, glob, json, os, sys
P, OUT, R, REPO = sys.argv[1:5]
sys.path.insert(0, REPO)
import bench
res = {}
for f in sorted(glob.glob(P + "/*.csv")):
    c = os.path.basename(f)[:-4]
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "height_scan" in r.get("Kernel_Name", "") and r.get("Counter_Name") == c]
    if vals:
        res[c] = sum(vals[1:]) / max(1, len(vals[1:]))
N, pts = 65536, 140
out = {"arenas": N, "points_per_arena": pts, "counters_per_launch": res, "kernel_src_sha256": bench.kernel_source_hash(),
       "what": "ois_height_scan at 65 536 arenas spread over the 6 x 2-tile field after 60 random steps (tools/height_scan_probe.py)"}
if "TCP_TCC_READ_REQ_sum" in res:
    out["l2_read_requests_per_point"] = res["TCP_TCC_READ_REQ_sum"] / (N * pts)
    out["l2_read_requests_per_arena"] = res["TCP_TCC_READ_REQ_sum"] / N
json.dump(out, open(f"{OUT}/{R}_height_scan_pmc.json", "w"), indent=1)
print(json.dumps(out))
PY
