set -x
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export SUSHI_BENCH_CACHE=/tmp/sushi_bench_cache
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline"
rm -rf gpurun_out/prof_fetch gpurun_out/prof_write
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof_fetch -o fetch -- $B > gpurun_out/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d gpurun_out/prof_write -o write -- $B > gpurun_out/write.log 2>&1
python tools/summarize_pmc.py gpurun_out/pmc_fw.csv $(find gpurun_out/prof_fetch gpurun_out/prof_write -name '*counter_collection.csv'); grep -E "kernel|mac|ifft|refine|tspec" gpurun_out/pmc_fw.csv
timeout 300 python -m pytest tests/test_load_gpu.py tests/test_distributed_gpu.py -m gpu -q -x -s 2>&1 | tail -8
python bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/q_base.json 2>/dev/null; python -c "
import json;d=json.load(open('gpurun_out/q_base.json'));print('base',round(d['ms_per_step'],2),d['roofline']['stage_ms'])"
