#!/bin/bash
# kernel-trace stats + PMC passes (each in its own run) for the bench; summaries copied to gpurun_out/
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r1}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_$TAG
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG/trace -o bench -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --accuracy-steps 0 --no-exact --no-clip --no-side-blocks > $R/gpurun_out/prof_$TAG.trace.log 2>&1
for C in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES"; do
  N=$(echo $C | tr ' ' '_')
  timeout 900 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/prof_$TAG/pmc_$N -o bench -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --accuracy-steps 0 --no-exact --no-clip --no-side-blocks > $R/gpurun_out/prof_$TAG.pmc_$N.log 2>&1
  echo "pmc $C rc=$?"
done
cd $R
python scripts/make_pmc_summary.py gpurun_out/prof_$TAG gpurun_out/${TAG}_pmc_summary.json | head -6
find gpurun_out/prof_$TAG -name "*.csv" | xargs ls -la | head -30
# keep the merged payload small: drop per-dispatch traces, keep the stats and the counter summaries
find gpurun_out/prof_$TAG -name "*kernel_trace.csv" -delete
find gpurun_out/prof_$TAG -name "*counter_collection.csv" -size +10M -delete
