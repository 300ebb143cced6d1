#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_t; mkdir -p $R/gpurun_out
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_t -o bench -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --accuracy-steps 0 --no-exact --no-clip > $R/gpurun_out/prof_t.log 2>&1
cd $R
python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/prof_t/**/*kernel_stats.csv',recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:28]:
    print(r['Name'][:80], r['Calls'], r['TotalDurationNs'], r['AverageNs'], r['MinNs'], r['MaxNs'])
PY
find gpurun_out/prof_t -name "*kernel_trace.csv" -size +30M -delete
