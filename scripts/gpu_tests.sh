#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd $R
timeout 1500 python -m pytest tests -q -m gpu --tb=line -rf "$@" > gpurun_out/gpu_tests.log 2>&1
grep -E "^/|passed|failed|^FAILED|Error" gpurun_out/gpu_tests.log | cut -c1-400 | head -60
