# BASELINE configs[4] as stated: the FFT line with oracle sample + cpu_baseline, HBM counters of it, and the im2col+MFMA GEMM
# formulation (the direct Toeplitz kernel) on the same workload with MFMA counters.
set -x
O=gpurun_out/r03cfg4
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export SUSHI_BENCH_CACHE=/tmp/sushi_bench_cache
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "segment_count or spectra or long_template or random_search or ragged" > $O/pytest_quick.log 2>&1; tail -2 $O/pytest_quick.log
timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --cpu-sample 8 > $O/bench_cfg2_quick.json 2>/dev/null; python -c "import json;d=json.load(open(\"$O/bench_cfg2_quick.json\"));print(d[\"value\"],d[\"roofline\"][\"stage_ms\"])"
rocprofv3 -L 2>/dev/null | grep -iE "mfma|SQ_BUSY|SQ_INSTS_VALU" | head -40 > $O/counters_avail.log; cat $O/counters_avail.log | head -30
timeout 900 python bench.py --config 4 --steps 5 --warmup 2 > $O/bench_cfg4_n1.json 2> $O/bench_cfg4_n1.err; head -c 250 $O/bench_cfg4_n1.json; echo; tail -2 $O/bench_cfg4_n1.err
B="python bench.py --config 4 --steps 2 --warmup 1 --no-cpu-baseline --cpu-sample 2"
rm -rf gpurun_out/prof4_*
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof4_kt -o kt -- $B > $O/kt.log 2>&1
timeout 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof4_fetch -o fetch -- $B > $O/fetch.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d gpurun_out/prof4_write -o write -- $B > $O/write.log 2>&1
cp gpurun_out/prof4_kt/kt_kernel_stats.csv $O/cfg4_kernel_stats.csv
python tools/summarize_pmc.py $O/cfg4_pmc_summary.csv $(find gpurun_out/prof4_fetch gpurun_out/prof4_write -name '*counter_collection.csv'); cat $O/cfg4_pmc_summary.csv
D="python bench.py --config 4 --path direct --steps 1 --warmup 0 --no-cpu-baseline --cpu-sample 16"
timeout 600 $D > $O/bench_cfg4_direct_n1.json 2> $O/bench_cfg4_direct_n1.err; head -c 250 $O/bench_cfg4_direct_n1.json; echo; tail -2 $O/bench_cfg4_direct_n1.err
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE --kernel-include-regex "match_sqdiff" --output-format csv -d gpurun_out/prof4_mfma -o m -- $D > $O/mfma.log 2>&1; tail -3 $O/mfma.log
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "match_sqdiff" --output-format csv -d gpurun_out/prof4_dfetch -o f -- $D > $O/dfetch.log 2>&1
python tools/summarize_pmc.py $O/cfg4_direct_pmc_summary.csv $(find gpurun_out/prof4_mfma gpurun_out/prof4_dfetch -name '*counter_collection.csv'); cat $O/cfg4_direct_pmc_summary.csv
