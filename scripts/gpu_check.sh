#!/bin/bash
# One gpurun call: GPU test-suite, smoke, default bench [, rocprof profile].  Logs under gpurun_out/.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
TAG=${1:-check}
timeout ${2:-1500} python -m pytest tests -q -m gpu --tb=short -rf --durations=8 > gpurun_out/${TAG}_tests.log 2>&1
echo "pytest rc=$?"; grep -E "passed|failed|^FAILED|^ERROR" gpurun_out/${TAG}_tests.log | cut -c1-300 | head -20
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
SECONDS=0; timeout 900 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; echo "bench rc=$? wall ${SECONDS}s"; cut -c1-600 gpurun_out/${TAG}_bench.json; tail -3 gpurun_out/${TAG}_bench.err
if [ "$3" = "prof" ]; then bash scripts/gpu_profile.sh $TAG 2>&1 | tail -12; fi
