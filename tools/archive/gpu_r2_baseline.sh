# Round-2 baseline on the MI355X box: GPU tests (incl. the reference-executed goldens), bench at the north-star
# configuration with the round-1 kernels, the hard workload, kernel trace + two PMC passes.
set -x
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export SUSHI_BENCH_CACHE=/tmp/sushi_bench_cache
timeout 900 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu.log 2>&1; tail -5 gpurun_out/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python bench.py --steps 20 --warmup 3 > gpurun_out/bench_cfg2.json 2> gpurun_out/bench_cfg2.err; cut -c1-1500 gpurun_out/bench_cfg2.json; tail -3 gpurun_out/bench_cfg2.err
timeout 600 python bench.py --hard-frac 0.05 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench_cfg2_hard.json 2> gpurun_out/bench_cfg2_hard.err; python -c "import json;d=json.load(open('gpurun_out/bench_cfg2_hard.json'));print('hard',d['value'],d['ms_per_step'],d['roofline']['stage_ms'],d['roofline']['searches_finished_by_fallback_kernel'],d['parity'])"; tail -3 gpurun_out/bench_cfg2_hard.err
B="python bench.py --steps 5 --warmup 1 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_kt -o kt -- $B > gpurun_out/kt.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof_fetch -o fetch -- $B > gpurun_out/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d gpurun_out/prof_write -o write -- $B > gpurun_out/write.log 2>&1
python tools/summarize_pmc.py gpurun_out/pmc_cfg2.csv $(find gpurun_out/prof_fetch gpurun_out/prof_write -name '*counter_collection.csv') ; cat gpurun_out/pmc_cfg2.csv
find gpurun_out/prof_kt -name '*kernel_stats.csv' -exec head -12 {} \;
