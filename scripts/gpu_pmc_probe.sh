#!/bin/bash
# PMC counters of the kernel probe (scripts/probe_kernels.py <what> <mode>): separate rocprofv3 passes, summary printed.
R=${GRAFT_REPO_ROOT:-$(pwd)}
WHAT=${1:-wgrad}; MODE=${2:-f16x2}
mkdir -p $R/gpurun_out/pmc_probe
cd /tmp && export TMPDIR=/tmp
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/pmc_probe/p$i -o o -- python $R/scripts/probe_kernels.py $WHAT $MODE > /dev/null 2>&1
done
cd $R
python - <<'PY'
import csv, glob, collections
agg=collections.defaultdict(lambda: collections.defaultdict(lambda:[0,0.0]))
for f in glob.glob('gpurun_out/pmc_probe/p*/o_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'].split('(')[0][:44]
        if 'conv_nn' in k or 'gemm_nt' in k or 'clip' in k:
            a=agg[k][r['Counter_Name']]; a[0]+=1; a[1]+=float(r['Counter_Value'])
for k,v in agg.items():
    print(k)
    for c,(n,s) in sorted(v.items()): print(f"   {c:28s} {s/n:14.4e}  (n={n})")
PY
rm -rf gpurun_out/pmc_probe
