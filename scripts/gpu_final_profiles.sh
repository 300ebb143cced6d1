#!/bin/bash
# final round-1 evidence: full gpu tests, smoke, bench (fp32 default + bf16 info line), rocprof stats + PMC
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -m gpu --tb=line -rf > gpurun_out/gpu_tests.log 2>&1
grep -E "^/|passed|failed|^FAILED|rror" gpurun_out/gpu_tests.log | cut -c1-300 | head -20
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python bench.py > gpurun_out/bench_f32.json 2> gpurun_out/bench_f32.err; cut -c1-250 gpurun_out/bench_f32.json
timeout 900 python bench.py --dtype bf16 --no-cpu-baseline > gpurun_out/bench_bf16.json 2>/dev/null; cut -c1-250 gpurun_out/bench_bf16.json
bash scripts/gpu_profile.sh r1final > /dev/null 2>&1
head -8 gpurun_out/prof_r1final/trace/bench_kernel_stats.csv | cut -c1-160
