#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd $R
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_exact_f32_gpu.py tests/test_bf16_gpu.py -q -m gpu --tb=short -x 2>&1 | tail -3
echo "== wide"; timeout 300 python scripts/probe_conv.py f32x3 2>&1 | grep conv
echo "== narrow"; BM_X3_WIDE=0 timeout 300 python scripts/probe_conv.py f32x3 f32 bf16 2>&1 | grep conv
timeout 900 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/bench.log 2>&1; tail -1 gpurun_out/bench.log | cut -c1-300
