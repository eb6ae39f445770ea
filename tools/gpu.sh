#!/bin/bash
# build locally (the box gets the built .so files with the snapshot), then run a command on an MI355X box
# usage: tools/gpu.sh <timeout-seconds> '<command>'
set -e
cd "$(dirname "$0")/.."
make -C structure-from-motion-3d-reconstruction_amd/csrc -j8 all 2>&1 | grep -E "error|Error" && exit 1
make -C oracle oracle >/dev/null
exec /usr/local/graft/bin/gpurun --timeout "$1" -- "$2"
