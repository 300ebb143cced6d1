#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_bf16_gpu.py -q -m gpu --tb=line -rf 2>&1 | grep -E "^/|passed|failed|^FAILED|rror" | cut -c1-300 | head -30
cd scripts && timeout 600 python perf_probe_bf16.py 2>&1 | tail -30
