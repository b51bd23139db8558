#!/bin/bash
# LDS bank conflicts per kernel over one encoder pass, one prefill and two eager decode steps (batch 32, 1.7B shapes):
# SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE.  Usage (on the GPU box): bash tools/pmc_lds.sh
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
rm -rf $O/pl1
timeout -k 10 400 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/pl1 -- python3 $R/tools/pmc_probe.py encoder,prefill,step > $O/pl1.log 2>&1 || { tail -5 $O/pl1.log; exit 1; }
python3 - <<'PY'
import csv, glob, os, collections
R=os.environ.get('GRAFT_REPO_ROOT','/root/repo')
f=glob.glob(f'{R}/gpurun_out/pl1/**/*counter_collection.csv', recursive=True)
acc=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for row in csv.DictReader(open(f[0])):
    k=row['Kernel_Name'][:60]
    acc[k][row['Counter_Name']]+=float(row['Counter_Value'])
    cnt[(k,row['Counter_Name'])]+=1
rows=[]
for k,v in acc.items():
    if v.get('SQ_LDS_IDX_ACTIVE',0) > 0:
        rows.append((v['SQ_LDS_BANK_CONFLICT']/v['SQ_LDS_IDX_ACTIVE'], k, int(v['SQ_LDS_IDX_ACTIVE']), int(v['SQ_LDS_BANK_CONFLICT']), int(v.get('SQ_BUSY_CYCLES',0))))
for r in sorted(rows, reverse=True)[:25]:
    print(f"{r[0]:6.1%}  idx_active {r[2]:>12}  conflict {r[3]:>12}  busy {r[4]:>10}  {r[1]}")
PY
find $O/pl1 -name '*.csv' -size +5M -delete
