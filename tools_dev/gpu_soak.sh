#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 300 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
for gname in Velodyne64E Velodyne32E VelodyneVLP16; do
  timeout 600 python3 tools_dev/soak_fullsize.py 1024 900000 $gname 2>&1 | tail -1
done
timeout 900 python3 tools_dev/soak_general.py 2048 950000 2>&1 | tail -1
