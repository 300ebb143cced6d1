R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out/pmc_clip
cd /tmp && export TMPDIR=/tmp
i=0
for C in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/pmc_clip/p$i -o o -- python $R/scripts/probe_kernels.py clip f16x2 > /dev/null 2>&1
done
cd $R
python - <<'PY'
import csv, glob, collections
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('gpurun_out/pmc_clip/p*/o_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'].split('(')[0][:50]
        if 'gemm_nt' in k or 'clip' in k or 'conv_nn' in k or 'cand' in k:
            agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in agg.items():
    print(k)
    for c,vals in sorted(v.items()):
        vals=sorted(vals)
        print(f"   {c:14s} n={len(vals):3d} min {vals[0]:12.4e} median {vals[len(vals)//2]:12.4e} max {vals[-1]:12.4e}")
PY
rm -rf gpurun_out/pmc_clip
