#!/bin/bash
# first GPU call: kernel parity + kernel timing probe
mkdir -p gpurun_out
rocminfo | grep -E "gfx|Compute Unit" | head -4 > gpurun_out/rocminfo.txt 2>&1
nproc > gpurun_out/nproc.txt
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu 2>&1 | tail -60 > gpurun_out/kernels_test.log
timeout 600 python scripts/perf_probe.py > gpurun_out/perf_probe.log 2>&1
tail -5 gpurun_out/kernels_test.log
cat gpurun_out/perf_probe.log
