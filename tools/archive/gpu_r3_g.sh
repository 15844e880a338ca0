set -x
O=gpurun_out/r03g
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export SUSHI_BENCH_CACHE=/tmp/sushi_bench_cache
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_cfg2_n1.json 2> $O/bench_cfg2_n1.err; head -c 250 $O/bench_cfg2_n1.json; echo; tail -2 $O/bench_cfg2_n1.err
timeout 120 python tools/stage_times.py --steps 10 --tag product 2>/dev/null | tail -1 | tee -a $O/ab.log
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_ccoeff.py -m gpu -q -x > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
