#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
timeout 900 python scripts/fullsize_parity.py cfg2 cfg3 2>&1 | tail -4
for w in cfg3 cfg5; do
  timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --workload $w 2>/dev/null | tee gpurun_out/bench_$w.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$w', round(d['ms_per_step'],2), 'ms', round(d['value']), 'seg/s')"
done
timeout 900 python bench.py --steps 20 --warmup 3 > gpurun_out/bench.log 2>&1; tail -1 gpurun_out/bench.log | cut -c1-300
