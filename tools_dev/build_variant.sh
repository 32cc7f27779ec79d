#!/bin/bash
# developer helper (build container): a library variant for an A/B run (tools_dev/ab.sh) without -D knobs in the product source:
# copies csrc/, applies the sed expressions, builds r-pcc_amd/lib/variants/<name>.so (git-ignored; travels with gpurun).
# usage: tools_dev/build_variant.sh <name> 's/#define BAND_WG_PER_XCD 32/#define BAND_WG_PER_XCD 16/' [more sed expressions]
set -e
cd "$(dirname "$0")/.."
name=$1; shift
d=$(mktemp -d /tmp/rpcc_var_XXXX)
mkdir -p $d/r-pcc_amd $d/include r-pcc_amd/lib/variants
cp -r r-pcc_amd/csrc $d/r-pcc_amd/; cp include/*.h $d/include/
for e in "$@"; do
  before=$(cat $d/r-pcc_amd/csrc/* | md5sum)
  sed -i -E "$e" $d/r-pcc_amd/csrc/*
  [ "$before" == "$(cat $d/r-pcc_amd/csrc/* | md5sum)" ] && { echo "sed expression changed nothing: $e"; exit 1; }
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fPIC -shared -Wno-unused-value $d/r-pcc_amd/csrc/rpcc_hip.hip -o r-pcc_amd/lib/variants/$name.so
rm -rf $d; echo r-pcc_amd/lib/variants/$name.so
