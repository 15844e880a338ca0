# bench at the north-star configuration (+ the hard workload, + configs[1]), kernel trace
set -x
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export SUSHI_BENCH_CACHE=/tmp/sushi_bench_cache
timeout 900 python bench.py --steps 20 --warmup 3 ${BENCH_ARGS:-} > gpurun_out/bench_cfg2.json 2> gpurun_out/bench_cfg2.err; python -c "
import json;d=json.load(open('gpurun_out/bench_cfg2.json'));r=d['roofline'];print('cfg2',d['value'],d['ms_per_step'],r['stage_ms'],r['frac'],r['step_frac'],d['parity'],d['cpu_baseline'])"; tail -3 gpurun_out/bench_cfg2.err
timeout 600 python bench.py --hard-frac 0.05 --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/bench_cfg2_hard.json 2> gpurun_out/bench_cfg2_hard.err; python -c "
import json;d=json.load(open('gpurun_out/bench_cfg2_hard.json'));print('hard',d['value'],d['ms_per_step'],d['roofline']['stage_ms'],d['roofline']['diagnostics'],d['parity'])"; tail -3 gpurun_out/bench_cfg2_hard.err
timeout 600 python bench.py --config 1 --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/bench_cfg1.json 2> gpurun_out/bench_cfg1.err; python -c "
import json;d=json.load(open('gpurun_out/bench_cfg1.json'));print('cfg1',d['value'],d['ms_per_step'],d['roofline']['stage_ms'])"
B="python bench.py --steps 5 --warmup 1 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_kt -o kt -- $B > gpurun_out/kt.log 2>&1
find gpurun_out/prof_kt -name '*kernel_stats.csv' -exec head -14 {} \;
