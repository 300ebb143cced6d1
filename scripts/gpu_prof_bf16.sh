#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_bf16
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_bf16 -o bench -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --dtype bf16 > $R/gpurun_out/prof_bf16.log 2>&1
cd $R
head -22 gpurun_out/prof_bf16/bench_kernel_stats.csv | cut -c1-150
rm -f gpurun_out/prof_bf16/bench_kernel_trace.csv
