#!/bin/bash
# developer helper (build container): the library of the LAST COMMIT as r-pcc_amd/lib/variants/head.so (git-ignored; travels with gpurun), so that
# tools_dev/ab_head.sh can alternate the working tree's library with it on the GPU box.
set -e
cd "$(dirname "$0")/.."
d=$(mktemp -d /tmp/rpcc_head_XXXX); mkdir -p r-pcc_amd/lib/variants
git archive HEAD r-pcc_amd/csrc include | tar -x -C $d
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fPIC -shared -Wno-unused-value $d/r-pcc_amd/csrc/rpcc_hip.hip -o r-pcc_amd/lib/variants/head.so
rm -rf $d; echo r-pcc_amd/lib/variants/head.so
