#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
python3 bench.py --input tests/golden/example_64E.npz 2>/dev/null | tail -1 | cut -c1-330
python3 bench.py --input tests/golden/example_64E.npz --config 2 2>/dev/null | tail -1 | cut -c1-330
python3 bench.py --input tests/golden/example_64E.npz --config 2 --pipeline 1 2>/dev/null | tail -1 | cut -c1-330
