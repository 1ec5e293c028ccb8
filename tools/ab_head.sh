#!/bin/bash
# A/B on one box: the step kernels of an earlier commit (built HERE with `tools/ab_build.sh <rev> <name>`) against HEAD's, twice
# each, alternating: launch intercept + per-substep slope (inloop_probe), the launch inside bench.py's actor loop, fused rollouts.
#   tools/ab_build.sh 597101e r3body && gpurun -- 'bash tools/ab_head.sh r3body > gpurun_out/r4_ab_head.txt'
set -e -o pipefail
NAME=${1:-r3body}
HASH=$(python3 -c "import bench; print(bench.kernel_source_hash())")
echo "# kernel sources sha256 $HASH (current); $NAME = tools/bin/libois_$NAME.so (tools/ab_build.sh: the step kernels of an earlier commit)"
pick='import json,sys; d=json.loads(sys.stdin.readline()); print("bench.py actor loop: ms_per_step %.4f, step kernel by events %.2f us" % (d["ms_per_step"], 1e3*d["roofline"]["kernel_ms"]))'
for v in $NAME current $NAME current; do
  export OIS_LIB_OVERRIDE=$PWD/tools/bin/libois_$v.so
  echo "== $v"
  python tools/inloop_probe.py 2>/dev/null
  python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-extras 2>/dev/null | python -c "$pick"
  python tools/launch_overhead_probe.py 65536 2>/dev/null
done
