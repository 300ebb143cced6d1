#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu --tb=short -x -k conv 2>&1 | tail -2
echo "== wide"; timeout 300 python scripts/probe_conv.py f32x3 2>&1 | grep conv
