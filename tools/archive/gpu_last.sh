set -x
mkdir -p gpurun_out/r02g
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export SUSHI_BENCH_CACHE=/tmp/sushi_bench_cache
O=gpurun_out/r02g
B="python bench.py --steps 5 --warmup 1 --no-cpu-baseline"
timeout 80 $B > $O/bench_cfg2_nocpu.json 2> $O/warm.err
timeout 70 python -m pytest tests/test_gpu_parity.py -m gpu -q -x > $O/pytest_parity.log 2>&1; tail -1 $O/pytest_parity.log
timeout 45 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof_fetch -o fetch -- $B > $O/fetch.log 2>&1
timeout 45 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d gpurun_out/prof_write -o write -- $B > $O/write.log 2>&1
python tools/summarize_pmc.py $O/cfg2_pmc_summary.csv $(find gpurun_out/prof_fetch gpurun_out/prof_write -name '*counter_collection.csv'); grep -E "mac_kernel|ifft_kernel" $O/cfg2_pmc_summary.csv
