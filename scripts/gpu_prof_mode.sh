#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
M=${1:-f32x3}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_$M
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$M -o bench -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --dtype $M > $R/gpurun_out/prof_$M.log 2>&1
cd $R
head -16 gpurun_out/prof_$M/bench_kernel_stats.csv | cut -c1-130
rm -f gpurun_out/prof_$M/bench_kernel_trace.csv
