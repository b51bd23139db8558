#!/bin/bash
# Wave-cycle anatomy of the kernels of one encoder pass + one prefill (+ two eager decode steps with `step`):
# bash tools/pmc_kernel.sh <kernel name substring> [probe list]   (on the GPU box)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
PAT=${1:-conv1}
WHAT=${2:-encoder}
rm -rf $O/pk1 $O/pk2
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES --kernel-trace --output-format csv -d $O/pk1 -- python3 $R/tools/pmc_probe.py $WHAT > $O/pk1.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU_TRANS SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU --kernel-trace --output-format csv -d $O/pk2 -- python3 $R/tools/pmc_probe.py $WHAT > $O/pk2.log 2>&1 || exit 1
python3 - "$PAT" <<'PY'
import csv, glob, os, sys, collections
R=os.environ.get('GRAFT_REPO_ROOT','/root/repo')
pat=sys.argv[1]
for d in ('pk1','pk2'):
    f=glob.glob(f'{R}/gpurun_out/{d}/**/*counter_collection.csv', recursive=True)
    if not f: print(d,'no counters (see', d+'.log)'); continue
    acc=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
    for row in csv.DictReader(open(f[0])):
        k=row['Kernel_Name'][:50]
        acc[k][row['Counter_Name']]+=float(row['Counter_Value']); cnt[(k,row['Counter_Name'])]+=1
    for k in acc:
        if pat in k:
            n=max(cnt[(k,c)] for c in acc[k])
            print(d,k,'dispatches',n,{c:round(v/n) for c,v in acc[k].items()})
PY
find $O/pk1 $O/pk2 -name '*.csv' -size +5M -delete
