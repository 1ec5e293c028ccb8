#!/bin/bash
# A/B of the two plane-kernel shells (step_kernel: compiler's register budget, no scratch;
# step_kernel_w2: held to 256 registers, two waves per SIMD) at 65 536 arenas, with the actor in
# the loop and env-only, by stream events (bench.py) and under rocprofv3 --kernel-trace --stats.
#   gpurun -- 'bash tools/waves_ab.sh > gpurun_out/waves_ab.txt'
set -e -o pipefail
export TMPDIR=/tmp
P=/tmp/ois_ab_$$; mkdir -p $P
pick='import json,sys; d=json.loads(sys.stdin.readline()); print("  value %.4g env-steps/s  ms_per_step %.4f  step kernel by events %.1f us" % (d["value"], d["ms_per_step"], 1e3*d["roofline"]["kernel_ms"]))'
for W in 1 2; do
  export OIS_STEP_WAVES=$W
  echo "== OIS_STEP_WAVES=$W (65 536 arenas)"
  echo " actor in the loop:"; python bench.py --steps 300 --warmup 20 --no-cpu-baseline | python -c "$pick"
  echo " env only:";          python bench.py --steps 400 --warmup 40 --no-cpu-baseline --no-actor | python -c "$pick"
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $P/kt$W -- python3 $OLDPWD/bench.py --steps 100 --warmup 10 --no-cpu-baseline > /dev/null 2> $P/kt$W.err) || { tail -5 $P/kt$W.err; exit 1; }
  echo " rocprofv3 --kernel-trace --stats, actor in the loop (name, calls, total ns, avg ns, %, min, max, stddev):"
  grep step_kernel $(find $P/kt$W -name '*kernel_stats.csv' | head -1) | sed 's/(ois::DevConsts.*)"/(...)"/' | cut -c1-160
done
