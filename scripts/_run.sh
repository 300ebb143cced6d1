#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}; mkdir -p gpurun_out; export TMPDIR=/tmp
rm -rf gpurun_out/prof_wg; 
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_wg -o wg -- python scripts/probe_kernels.py wgrad f16x2 > /dev/null 2>&1
python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/prof_wg/**/*kernel_stats.csv',recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:12]:
    print(r['Name'][:70], r['Calls'], r['AverageNs'], r['MinNs'], r['MaxNs'])
PY
