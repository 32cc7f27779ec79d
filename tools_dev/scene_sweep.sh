#!/bin/bash
# GPU box: the tile-pruned FPS against the brute-force kernel on the headline scene and three adversarial ones
# -> gpurun_out/scene_sweep.txt   (serial steps: the FPS launch time is the kernel's own; plus the pipelined frames/s)
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
out=gpurun_out/scene_sweep.txt
echo "| scene | FPS kernel | FPS launch ms (serial steps) | ms per step serial | frames/s, 3 batches in flight | verified |" > $out
echo "|---|---|---|---|---|---|" >> $out
for scene in default shell noise corridor; do for mode in pruned brute; do
  fl=""; [ $mode == brute ] && fl="--fps-bruteforce"
  a=$(timeout 600 python3 bench.py --no-secondary --cpu-sample 0 --verify-frames 8 --scene $scene $fl --pipeline 1 --steps 10 --warmup 2 2>/dev/null | tail -1)
  b=$(timeout 600 python3 bench.py --no-secondary --cpu-sample 0 --verify-frames 8 --scene $scene $fl --steps 20 --warmup 3 2>/dev/null | tail -1)
  python3 - "$scene" "$mode" "$a" "$b" >> $out <<'PY'
import sys, json
s, m, a, b = sys.argv[1:5]
a, b = json.loads(a), json.loads(b)
print("| %s | %s | %.3f | %.3f | %.0f | %s / %s |" % (s, m, a["roofline"]["dominant_kernel"]["launch_ms"], a["ms_per_step"], b["value"], a["verified"], b["verified"]))
PY
done; done
cat $out
