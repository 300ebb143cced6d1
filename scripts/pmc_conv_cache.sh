#!/bin/bash
# L1 / L2 request counters of the wide conv (scripts/probe_kernels.py conv f16x2) for the two main loops: how many of the
# register-fed loop's A-fragment loads (each fragment is fetched by the two wavefronts of a row pair) are served by the L1.
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out/pmc_cc
cd /tmp && export TMPDIR=/tmp
for V in 0 1; do
  i=0
  for C in "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_READ_sum"; do
    i=$((i+1))
    BM_CONV_LDSDMA=$V timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/pmc_cc/v${V}_p$i -o o -- python $R/scripts/probe_kernels.py conv f16x2 > /dev/null 2>&1
  done
done
cd $R
python - <<'PY'
import csv, glob, collections
for V in (0, 1):
    agg=collections.defaultdict(lambda: collections.defaultdict(lambda:[0,0.0]))
    for f in glob.glob(f'gpurun_out/pmc_cc/v{V}_p*/o_counter_collection.csv'):
        for r in csv.DictReader(open(f)):
            k=r['Kernel_Name'].split('(')[0][:48]
            if 'conv_nn_h2' in k:
                a=agg[k][r['Counter_Name']]; a[0]+=1; a[1]+=float(r['Counter_Value'])
    print("BM_CONV_LDSDMA =", V)
    for k,v in agg.items():
        print("  ", k, "  ".join(f"{c}={s/n:.4e}" for c,(n,s) in sorted(v.items())))
PY
rm -rf gpurun_out/pmc_cc
