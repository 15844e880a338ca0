# parity tests, bench, one PMC pass (WRITE_SIZE / L2 hits) of the default workload
set -x
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export SUSHI_BENCH_CACHE=/tmp/sushi_bench_cache
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x > gpurun_out/pytest_quick.log 2>&1; tail -3 gpurun_out/pytest_quick.log
python bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/q_base.json 2>gpurun_out/q.err; python -c "
import json;d=json.load(open('gpurun_out/q_base.json'));print('base',round(d['value']),round(d['ms_per_step'],2),{k:round(v,2) for k,v in d['roofline']['stage_ms'].items()})"; tail -2 gpurun_out/q.err
rm -rf gpurun_out/prof_write
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d gpurun_out/prof_write -o write -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/write.log 2>&1
python tools/summarize_pmc.py gpurun_out/pmc_w.csv $(find gpurun_out/prof_write -name '*counter_collection.csv'); grep -E "kernel|mac|ifft" gpurun_out/pmc_w.csv
