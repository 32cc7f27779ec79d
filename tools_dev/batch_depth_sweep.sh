#!/bin/bash
# GPU box: frames per call x calls in flight at (roughly) constant frames in flight and beyond
# usage: tools_dev/batch_depth_sweep.sh            -> gpurun_out/batch_depth_sweep.txt
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
out=gpurun_out/batch_depth_sweep.txt; : > $out
run() {  # batch depth
  steps=$(( 100 * 256 / $1 ))
  r=$(timeout 300 python3 bench.py --no-secondary --cpu-sample 0 --verify-frames 4 --batch $1 --pipeline $2 --steps $steps --warmup $(( 3 * $2 )) 2>/dev/null | tail -1)
  echo "$1 $2 $r" | python3 -c "
import sys, json
b, d, rest = sys.stdin.read().split(' ', 2)
r = json.loads(rest)
print('| %4s | %2s | %8.0f | %.4f | %.4f | %s | %.3f |' % (b, d, r['value'], r['ms_per_step'], r['ms_per_step'] * 256 / int(b), r['verified'], r['roofline']['dominant_kernel']['launch_ms']))" >> $out
}
echo "| frames per call | calls in flight | frames/s | ms per call | ms per 256 frames | verified | FPS launch ms |" >> $out
echo "|---|---|---|---|---|---|---|" >> $out
for rep in 1 2; do
run 256 3
for cfg in "64 12" "64 8" "64 6" "128 6" "128 4" "128 3" "256 2" "256 4" "256 6" "512 1" "512 2" "512 3" "1024 1" "1024 2"; do run $cfg; done
done
run 256 3
cat $out
