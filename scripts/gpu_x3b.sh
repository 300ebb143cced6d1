#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_x3_gpu.py -q -m gpu --tb=line -rf -x -k "conv or error" 2>&1 | grep -E "^/|passed|failed|^FAILED|rror" | cut -c1-300 | head -10
for d in f32x3 f32x3; do
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --dtype $d 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$d', round(d['ms_per_step'],2), 'ms', round(d['value']), 'seg/s', d['roofline']['kernel'], round(d['roofline']['achieved'],1), 'TF', 'loss', d['config']['final_loss'])"
done
