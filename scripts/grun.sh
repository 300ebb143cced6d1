#!/bin/bash
# Local helper (build container): make sure the in-tree libbmhip.so is newer than every source and exports every
# declared symbol, then hand the command to gpurun.  Usage: scripts/grun.sh <timeout-seconds> '<command>'
set -e
cd "$(dirname "$0")/.."
python - <<'PY'
from brainmagick_amd import _lib
path = _lib.build_library(verbose=False)
_lib.lib()      # raises if a declared symbol is missing
print("libbmhip up to date:", path)
PY
exec /usr/local/graft/bin/gpurun --timeout "$1" -- "$2"
