cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
rm -rf $O/pa1 $O/pa2
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $O/pa1 -- python3 $R/tools/pmc_probe.py prefill > $O/pa1.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES --kernel-trace --output-format csv -d $O/pa2 -- python3 $R/tools/pmc_probe.py prefill > $O/pa2.log 2>&1 || exit 1
python3 - <<'PY'
import csv, glob, os, collections
R=os.environ.get('GRAFT_REPO_ROOT','/root/repo')
for d in ('pa1','pa2'):
    f=glob.glob(f'{R}/gpurun_out/{d}/**/*counter_collection.csv', recursive=True)
    if not f: print(d,'no file'); continue
    acc=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
    for row in csv.DictReader(open(f[0])):
        k=row['Kernel_Name'][:40]
        acc[k][row['Counter_Name']]+=float(row['Counter_Value'])
        cnt[(k,row['Counter_Name'])]+=1
    for k in acc:
        if 'attn_fwd' in k or 'gemm8p' in k:
            n=max(cnt[(k,c)] for c in acc[k])
            print(d,k,'dispatches',n,{c:round(v/n) for c,v in acc[k].items()})
PY
find $O/pa1 $O/pa2 -name '*.csv' -size +5M -delete
