#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd $R
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -120 > gpurun_out/gpu_tests.log
grep -E "^E  |passed|failed|^FAILED" gpurun_out/gpu_tests.log | head -60
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; tail -3 gpurun_out/smoke.log
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_r1
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r1 -o bench -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $R/gpurun_out/rocprof.log 2>&1
cd $R
tail -2 gpurun_out/rocprof.log
find gpurun_out/prof_r1 -type f | head
find gpurun_out/prof_r1 -name "*kernel_stats*" | head -1 | xargs -r head -40
