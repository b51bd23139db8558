#!/bin/bash
# HBM traffic (FETCH_SIZE x 2, WRITE_SIZE) of every kernel of an eager batch-32 decode step: two rocprofv3 --pmc passes over
# tools/pmc_probe.py step, summarised by tools/pmc_summary.py into gpurun_out/st_pmc_summary.json.  Run on the GPU box.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $O/st_fetch $O/st_write
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/st_fetch -- python3 $R/tools/pmc_probe.py step > $O/st_fetch.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/st_write -- python3 $R/tools/pmc_probe.py step > $O/st_write.log 2>&1 || exit 1
python3 $R/tools/pmc_summary.py $(find $O/st_fetch -name '*counter_collection.csv') $(find $O/st_write -name '*counter_collection.csv') $O/st_pmc_summary.json $O/pmc_probe_meta.json 2>&1 | grep -E "attn_decode|resnorm|embed_norm|sample|skinny4" | cut -c1-260
find $O/st_fetch $O/st_write -name '*.csv' -size +20M -delete
