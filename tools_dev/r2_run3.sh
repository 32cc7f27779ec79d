#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
python3 -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1
timeout 1200 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q -m gpu -k "fps or fused or golden or bench_configuration or full_size or fuzz" > gpurun_out/t3.log 2>&1; echo "rc=$?" >> gpurun_out/t3.log
tail -5 gpurun_out/t3.log
bash tools_dev/r2_variants.sh "" "-DFPS_TT_BATCH=1024" "-DFPS_TT_BATCH=256" "-DFPS_NO_REGTAB"
