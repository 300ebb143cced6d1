#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
for i in 1 2; do
  for v in 0 1; do
    BM_STATS_IN_EPILOGUE=$v timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('epilogue_stats=$v', round(d['ms_per_step'],2), 'ms', round(d['value']), 'seg/s', 'conv5', round(d['roofline']['achieved'],1))"
  done
done
