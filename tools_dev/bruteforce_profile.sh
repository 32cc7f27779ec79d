#!/bin/bash
# GPU box: the brute-force FPS kernel (the reference algorithm's stream-once case) under rocprofv3: kernel stats + HBM-side traffic
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
mkdir -p gpurun_out; rm -rf /tmp/bf && mkdir -p /tmp/bf
python3 bench.py --no-secondary --fps-bruteforce --pipeline 1 --cpu-sample 0 --steps 5 --warmup 1 > gpurun_out/bf_bench.json 2>/tmp/bf/e0
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/bf/s -o p -- python3 bench.py --no-secondary --fps-bruteforce --pipeline 1 --cpu-sample 0 --steps 5 --warmup 1 > /tmp/bf/b1 2>/tmp/bf/e1
cp $(find /tmp/bf/s -name "*kernel_stats.csv" | head -1) gpurun_out/bf_kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/bf/$c -o p -- python3 bench.py --no-secondary --fps-bruteforce --pipeline 1 --cpu-sample 0 --steps 2 --warmup 1 > /tmp/bf/$c.log 2>&1
  grep -E "Counter_Name|fps_range_kernel" $(find /tmp/bf/$c -name "*counter_collection.csv" | head -1) > gpurun_out/bf_pmc_$c.csv
done
tail -c 900 gpurun_out/bf_bench.json; grep fps_range gpurun_out/bf_kernel_stats.csv | cut -c1-200
