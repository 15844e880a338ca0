# End-of-round evidence within a small GPU budget: every command under its own timeout (a rocprofv3 that aborts can
# hang in its signal handler until the box's limit).  PMC passes first, so that the bench line carries roofline.traffic.
set -x
mkdir -p gpurun_out/r02f
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export SUSHI_BENCH_CACHE=/tmp/sushi_bench_cache
O=gpurun_out/r02f
B="python bench.py --steps 5 --warmup 1 --no-cpu-baseline"
timeout 200 $B > $O/warm.json 2> $O/warm.err            # builds the stream cache
rm -rf gpurun_out/prof_*
timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_kt -o kt -- $B > $O/kt.log 2>&1
timeout 150 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof_fetch -o fetch -- $B > $O/fetch.log 2>&1
timeout 150 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d gpurun_out/prof_write -o write -- $B > $O/write.log 2>&1
cp gpurun_out/prof_kt/kt_kernel_stats.csv $O/cfg2_kernel_stats.csv
python tools/summarize_pmc.py $O/cfg2_pmc_summary.csv $(find gpurun_out/prof_fetch gpurun_out/prof_write -name '*counter_collection.csv')
python tools/make_pmc_traffic.py $O/cfg2_pmc_summary.csv profiles/pmc_traffic.json "config2/fft/float32/3000/w120/m120/n1" $COMMIT > /dev/null; cp profiles/pmc_traffic.json $O/pmc_traffic.json
timeout 300 python bench.py --steps 20 --warmup 5 > $O/bench_cfg2_n1.json 2> $O/bench_cfg2_n1.err; head -c 400 $O/bench_cfg2_n1.json; echo
timeout 400 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; tail -2 $O/pytest_gpu.log
timeout 60 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 400 python bench.py --config 4 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_cfg4_n1.json 2> $O/bench_cfg4_n1.err; head -c 300 $O/bench_cfg4_n1.json; echo
head -5 $O/cfg2_kernel_stats.csv
