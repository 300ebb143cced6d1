#!/bin/bash
# Round-6 development call: a pytest subset + interleaved A/B of the conv main loops on one box.
# Usage: scripts/r6_call.sh <tag> "<pytest -k expression or empty>" [ab]
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
TAG=$1
if [ -n "$2" ]; then
  timeout 1200 python -m pytest tests -q -m gpu --tb=short -rf -x -k "$2" > gpurun_out/${TAG}_tests.log 2>&1
  echo "pytest rc=$?"; grep -E "passed|failed|^FAILED|^ERROR|Error" gpurun_out/${TAG}_tests.log | cut -c1-300 | head -20
fi
if [ "$3" = "ab" ]; then
  bash scripts/ab_bench.sh $TAG "BM_CONV_LDSDMA=1" "BM_CONV_LDSDMA=0" "BM_CONV_LDSDMA=1" "BM_CONV_LDSDMA=0"
fi
