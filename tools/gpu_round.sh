#!/bin/bash
# One gpurun call = a list of steps, each bounded by its own timeout; a step that is killed (timeout, signal) ends the call
# (no GPU step is started after a killed one).  Usage: tools/gpu_round.sh <tag> step [step ...]
#   tests | tests_parity | bench | pmc | mfma | stats | probe_batch1 | probe_graph | probe_graph4
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
TAG=$1; shift
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
run() {   # run <seconds> <logfile> cmd...
    local secs=$1 log=$2; shift 2
    echo "== $(date +%T) $*" | tee -a $O/${TAG}_steps.log
    timeout -k 10 $secs "$@" > $log 2>&1
    local rc=$?
    echo "   rc=$rc" | tee -a $O/${TAG}_steps.log
    if [ $rc -eq 124 ] || [ $rc -ge 128 ]; then echo "step killed (rc=$rc): stopping this call" | tee -a $O/${TAG}_steps.log; tail -30 $log; exit $rc; fi
    return 0
}
prof() {  # prof <name> <probe list> <rocprofv3 args...>
    local name=$1 what=$2; shift 2
    rm -rf $O/${TAG}_$name
    run 420 $O/${TAG}_$name.log rocprofv3 "$@" --output-format csv -d $O/${TAG}_$name -- python3 $R/tools/pmc_probe.py $what
    find $O/${TAG}_$name -name '*.csv' -size +20M -delete          # keep the merge-back under the 64 MiB cap
}
for step in "$@"; do
    case $step in
    tests)        run 900 $O/${TAG}_tests.log python3 -m pytest $R/tests -m gpu -q -x --ignore $R/tests/test_parity_gpu.py --ignore $(realpath $R)/tests/test_parity_gpu.py ; tail -5 $O/${TAG}_tests.log ;;
    tests_parity) run 1100 $O/${TAG}_parity.log python3 -m pytest $R/tests/test_parity_gpu.py -m gpu -q -s ; grep -v '^\[parity\]' $O/${TAG}_parity.log | tail -40 ;;
    tests_k)      run 600 $O/${TAG}_tests_k.log python3 -m pytest $R/tests/test_kernels_gpu.py -m gpu -q -x ; tail -5 $O/${TAG}_tests_k.log ;;
    tests_all)    run 1100 $O/${TAG}_tests_all.log python3 -m pytest $R/tests -m gpu -q -x ; tail -5 $O/${TAG}_tests_all.log ;;
    bench)        run 600 $O/${TAG}_bench.err python3 $R/bench.py --steps 20 --warmup 3 ; grep '^{' $O/${TAG}_bench.err > $O/${TAG}_bench.json; grep -v '^{' $O/${TAG}_bench.err | tail -25 ;;
    benchq)       run 400 $O/${TAG}_benchq.err python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-pipeline --no-extra-configs ; grep '^{' $O/${TAG}_benchq.err > $O/${TAG}_benchq.json; grep -v '^{' $O/${TAG}_benchq.err | tail -12 ;;
    benchf8)      run 400 $O/${TAG}_benchf8.err python3 $R/bench.py --steps 5 --warmup 2 --weights fp8 --no-cpu-baseline --no-pipeline --no-extra-configs ; grep '^{' $O/${TAG}_benchf8.err > $O/${TAG}_benchf8.json; grep -v '^{' $O/${TAG}_benchf8.err | tail -8
                  run 400 $O/${TAG}_benchf8w.err python3 $R/bench.py --steps 5 --warmup 2 --weights fp8w --no-cpu-baseline --no-pipeline --no-extra-configs ; grep '^{' $O/${TAG}_benchf8w.err > $O/${TAG}_benchf8w.json; grep -v '^{' $O/${TAG}_benchf8w.err | tail -8 ;;
    rehearse2)    Q3ASR_BENCH_REHEARSE=1 run 500 $O/${TAG}_rehearse2.err python3 $R/bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline ; grep '^{' $O/${TAG}_rehearse2.err > $O/${TAG}_rehearse2.json; grep -v '^{' $O/${TAG}_rehearse2.err | tail -12 ;;
    config5)      run 400 $O/${TAG}_config5.log python3 $R/tools/run_config5.py fp8 20 ; cat $O/${TAG}_config5.log ;;
    pmc)          prof pmc_fetch decode,encoder,prefill --pmc FETCH_SIZE --kernel-trace
                  prof pmc_write decode,encoder,prefill --pmc WRITE_SIZE --kernel-trace
                  python3 $R/tools/pmc_summary.py $(find $O/${TAG}_pmc_fetch -name '*counter_collection.csv') $(find $O/${TAG}_pmc_write -name '*counter_collection.csv') $O/${TAG}_pmc_summary.json $O/pmc_probe_meta.json > $O/${TAG}_pmc_print.txt 2>&1; tail -20 $O/${TAG}_pmc_print.txt ;;
    mfma)         prof mfma encoder,prefill --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace
                  python3 $R/tools/mfma_util.py $(find $O/${TAG}_mfma -name '*counter_collection.csv') $(find $O/${TAG}_mfma -name '*kernel_trace.csv') $O/${TAG}_mfma_util.json > $O/${TAG}_mfma_print.txt 2>&1; tail -16 $O/${TAG}_mfma_print.txt ;;
    stats)        prof stats decode,encoder,prefill,step --kernel-trace --stats
                  find $O/${TAG}_stats -name '*kernel_stats.csv' -exec head -30 {} \; ;;
    trace_graph)  rm -rf $O/${TAG}_trace_graph; run 420 $O/${TAG}_trace_graph.log rocprofv3 --kernel-trace --output-format csv -d $O/${TAG}_trace_graph -- python3 $R/tools/pmc_probe.py graph4
                  python3 $R/tools/agg_trace.py $(find $O/${TAG}_trace_graph -name '*kernel_trace.csv') 24 > $O/${TAG}_trace_graph.txt; cat $O/${TAG}_trace_graph.txt
                  python3 $R/tools/in_situ.py $(find $O/${TAG}_trace_graph -name '*kernel_trace.csv') $O/pmc_probe_meta.json $O/${TAG}_decode_in_situ.json
                  find $O/${TAG}_trace_graph -name '*.csv' -size +20M -delete ;;
    trace_graph1) rm -rf $O/${TAG}_trace_graph1; run 420 $O/${TAG}_trace_graph1.log rocprofv3 --kernel-trace --output-format csv -d $O/${TAG}_trace_graph1 -- python3 $R/tools/pmc_probe.py graph4 --batch 1
                  python3 $R/tools/agg_trace.py $(find $O/${TAG}_trace_graph1 -name '*kernel_trace.csv') 24 > $O/${TAG}_trace_graph1.txt; cat $O/${TAG}_trace_graph1.txt; find $O/${TAG}_trace_graph1 -name '*.csv' -size +20M -delete ;;
    ab_chains)    run 400 $O/${TAG}_ab_chains.log python3 $R/tools/ab_chains.py ${AB_CHAINS_ARGS:-32 30 128 6 1,2,4} ; grep chains $O/${TAG}_ab_chains.log ;;
    tests_s)      run 600 $O/${TAG}_tests_s.log python3 -m pytest $R/tests/test_sampling_gpu.py $R/tests/test_e2e_gpu.py -m gpu -q -x ; tail -5 $O/${TAG}_tests_s.log ;;
    tests_fp8)    run 900 $O/${TAG}_fp8.log python3 -m pytest $R/tests/test_parity_gpu.py -m gpu -q -s -k "fp8_mfma or one_full_size_layer" ; grep -v '^\[parity\]' $O/${TAG}_fp8.log | tail -30 ;;
    bench_gemm)   run 400 $O/${TAG}_bench_gemm.log python3 $R/tools/bench_gemm.py ${BENCH_GEMM_ARGS:-5 8,2} ; cat $O/${TAG}_bench_gemm.log ;;
    longform)     run 500 $O/${TAG}_longform.log python3 $R/tools/longform_variants.py ${LONGFORM_ARGS:-fp8 128} ; grep -v amdgpu.ids $O/${TAG}_longform.log ;;
    ab_height)    run 500 $O/${TAG}_ab_height.log python3 $R/tools/ab_gemm_height.py ; grep -v amdgpu.ids $O/${TAG}_ab_height.log ;;
    tests_gemm)   run 600 $O/${TAG}_tests_gemm.log python3 -m pytest $R/tests/test_kernels_gpu.py -m gpu -q -x -k gemm ; tail -8 $O/${TAG}_tests_gemm.log ;;
    admit_cost)   run 300 $O/${TAG}_admit_cost.log python3 $R/tools/admit_cost.py ; grep "k=" $O/${TAG}_admit_cost.log ;;
    attn_stamps)  run 300 $O/${TAG}_attn_stamps.log python3 $R/tools/attn_stamps.py ; grep -v amdgpu.ids $O/${TAG}_attn_stamps.log ;;
    e2e_probe)    run 400 $O/${TAG}_e2e_probe.log python3 $R/tools/e2e_probe.py ${E2E_PROBE_ARGS:-8} ; grep -v amdgpu.ids $O/${TAG}_e2e_probe.log ;;
    ab_step)      run 500 $O/${TAG}_ab_step.log python3 $R/tools/ab_step.py ${AB_STEP_ARGS:-32 one=5:0,two=5:1 9} ; grep -v amdgpu.ids $O/${TAG}_ab_step.log | tail -12 ;;
    b1_front)     run 300 $O/${TAG}_b1_front.log python3 $R/tools/b1_front.py ${B1_FRONT_ARGS:-1 30} ; grep -v amdgpu.ids $O/${TAG}_b1_front.log ;;
    ab_epi)       run 500 $O/${TAG}_ab_epi.log python3 $R/tools/ab_gemm_epilogue.py ; grep -v amdgpu.ids $O/${TAG}_ab_epi.log ;;
    stats5)       rm -rf $O/${TAG}_stats5; run 420 $O/${TAG}_stats5.log rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_stats5 -- python3 $R/tools/config5_probe.py
                  find $O/${TAG}_stats5 -name '*kernel_trace.csv' -delete; find $O/${TAG}_stats5 -name '*kernel_stats.csv' -exec head -40 {} \; ; grep config5_probe $O/${TAG}_stats5.log ;;
    ab_sampler)   run 300 $O/${TAG}_ab_sampler.log python3 $R/tools/ab_sampler.py ${AB_SAMPLER_ARGS:-40 5 40} ; grep -v amdgpu.ids $O/${TAG}_ab_sampler.log ;;
    ab_attn)      run 300 $O/${TAG}_ab_attn.log python3 $R/tools/ab_attn.py ${AB_ATTN_ARGS:-3} ; grep -v amdgpu.ids $O/${TAG}_ab_attn.log ;;
    tests_attn)   run 600 $O/${TAG}_tests_attn.log python3 -m pytest $R/tests/test_kernels_gpu.py -m gpu -q -x -k attn ; tail -5 $O/${TAG}_tests_attn.log ;;
    serve)        run 500 $O/${TAG}_serve32.log python3 $R/tools/bench_serve.py 32 25,60,90,120 6 ; tail -1 $O/${TAG}_serve32.log > $O/${TAG}_serve32.json; grep offered $O/${TAG}_serve32.log | cut -c1-200
                  run 400 $O/${TAG}_serve64.log python3 $R/tools/bench_serve.py 64 90,130 6 ; tail -1 $O/${TAG}_serve64.log > $O/${TAG}_serve64.json; grep offered $O/${TAG}_serve64.log | cut -c1-200 ;;
    gemm_small)   run 300 $O/${TAG}_gemm_small.log python3 $R/tools/bench_gemm_small.py ; grep "k=" $O/${TAG}_gemm_small.log ;;
    finite)       run 300 $O/${TAG}_finite.log python3 $R/tools/pmc_probe.py finite,step,graph4,batch1 ; grep probe $O/${TAG}_finite.log | tail -12 ;;
    probe_batch1) prof probe_batch1 batch1 --pmc FETCH_SIZE --kernel-trace ;;
    probe_graph)  prof probe_graph graph --pmc FETCH_SIZE --kernel-trace ;;
    probe_graph4) prof probe_graph4 graph4 --pmc FETCH_SIZE --kernel-trace ;;
    probe_regraph) prof probe_regraph regraph --pmc FETCH_SIZE --kernel-trace ;;
    bench_stats)  rm -rf $O/${TAG}_bench_stats; run 600 $O/${TAG}_bench_stats.log rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_bench_stats -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-pipeline --no-extra-configs
                  python3 $R/tools/request_gaps.py $(find $O/${TAG}_bench_stats -name '*kernel_trace.csv') > $O/${TAG}_request_gaps.txt 2>&1; cat $O/${TAG}_request_gaps.txt
                  find $O/${TAG}_bench_stats -name '*kernel_trace.csv' -delete; find $O/${TAG}_bench_stats -name '*.csv' -size +20M -delete
                  find $O/${TAG}_bench_stats -name '*kernel_stats.csv' -exec head -16 {} \; ; grep '^{' $O/${TAG}_bench_stats.log | tail -1 | cut -c1-300 ;;
    bench_pmc)    echo "bench_pmc: refused -- the full bench under rocprofv3 --pmc aborts in the profiler after ~3.3e5 dispatch records (DESIGN.md section 2); use bench_pmc16 or tools/pmc_step.sh" ;;
    stamps)       run 300 $O/${TAG}_stamps.log python3 $R/tools/gemm_stamps.py ; cat $O/${TAG}_stamps.log ;;
    gemmk)        run 300 $O/${TAG}_gemmk.log python3 $R/tools/bench_gemm_k.py 5 ; cat $O/${TAG}_gemmk.log
                  ;;
    bench_pmc16)  rm -rf $O/${TAG}_bench_pmc16; run 600 $O/${TAG}_bench_pmc16.log rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/${TAG}_bench_pmc16 -- python3 $R/bench.py --steps 1 --warmup 1 --ndec 16 --no-cpu-baseline --no-pipeline --no-extra-configs
                  find $O/${TAG}_bench_pmc16 -name '*.csv' -size +1M -delete; tail -12 $O/${TAG}_bench_pmc16.log ;;
    stage)        run 300 $O/${TAG}_stage.log python3 $R/tools/stage_profile.py 32 30 $O/${TAG}_stage.json ; cat $O/${TAG}_stage.log ;;
    *) echo "unknown step $step" ;;
    esac
done
echo "== done" | tee -a $O/${TAG}_steps.log
