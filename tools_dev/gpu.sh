#!/bin/bash
# build everything here (the .so files travel with the snapshot), then run the command on a GPU box: tools_dev/gpu.sh <timeout s> '<command>'
set -e
cd "$(dirname "$0")/.."
python -c "import __graft_entry__ as g; g.build()"
exec /usr/local/graft/bin/gpurun --timeout "$1" -- "$2"
