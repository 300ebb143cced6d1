#!/bin/bash
# final round evidence: full gpu tests, smoke, rocprof stats + PMC summary, then bench (f32x3 default + bf16 info line)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -m gpu --tb=line -rf > gpurun_out/gpu_tests.log 2>&1
grep -E "^/|passed|failed|^FAILED|rror" gpurun_out/gpu_tests.log | cut -c1-300 | head -20
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
bash scripts/gpu_profile.sh r1final 2>&1 | tail -12
cp gpurun_out/r1final_pmc_summary.json profiles/r1_pmc_summary.json     # so that the bench below reports `traffic`
timeout 900 python bench.py > gpurun_out/bench_f32.json 2> gpurun_out/bench_f32.err; cut -c1-250 gpurun_out/bench_f32.json
timeout 900 python bench.py --dtype bf16 --no-cpu-baseline > gpurun_out/bench_bf16.json 2>/dev/null; cut -c1-250 gpurun_out/bench_bf16.json
head -8 gpurun_out/prof_r1final/trace/bench_kernel_stats.csv | cut -c1-160
