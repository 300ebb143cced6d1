#!/bin/bash
# second GPU call: full gpu test-suite, smoke, bench, rocprof kernel trace of the bench
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd $R
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -80 > gpurun_out/gpu_tests.log
tail -15 gpurun_out/gpu_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; tail -3 gpurun_out/smoke.log
timeout 900 python bench.py --steps 10 --warmup 2 > gpurun_out/bench.log 2>&1; tail -2 gpurun_out/bench.log
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r1 -o bench -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $R/gpurun_out/rocprof.log 2>&1
cd $R
ls -R gpurun_out/prof_r1 | head -20
find gpurun_out/prof_r1 -name "*kernel_stats*" | head -1 | xargs -r head -30
