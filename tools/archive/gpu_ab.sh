# tests + bench of the product library, bench of a variant library (A/B), kernel trace of the product
set -x
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export SUSHI_BENCH_CACHE=/tmp/sushi_bench_cache
timeout 1200 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu.log 2>&1; tail -5 gpurun_out/pytest_gpu.log
show() { python -c "
import json,sys;d=json.load(open(sys.argv[1]));r=d['roofline'];print(sys.argv[2],round(d['value']),round(d['ms_per_step'],2),{k:round(v,2) for k,v in r['stage_ms'].items()},round(r['frac'],3),r.get('diagnostics'),d['parity']['max_idx_err_vs_oracle_sample'])" $1 $2; }
timeout 900 python bench.py --steps 20 --warmup 3 > gpurun_out/bench_cfg2.json 2> gpurun_out/bench_cfg2.err; show gpurun_out/bench_cfg2.json cfg2; tail -2 gpurun_out/bench_cfg2.err
for v in ${VARIANTS:-}; do
  SUSHI_HIP_LIB=$GRAFT_REPO_ROOT/sushi_amd/lib/libsushi_hip_$v.so timeout 900 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/bench_cfg2_$v.json 2> gpurun_out/bench_cfg2_$v.err; show gpurun_out/bench_cfg2_$v.json cfg2_$v; tail -2 gpurun_out/bench_cfg2_$v.err
done
timeout 600 python bench.py --hard-frac 0.05 --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/bench_cfg2_hard.json 2> gpurun_out/bench_cfg2_hard.err; show gpurun_out/bench_cfg2_hard.json hard
timeout 600 python bench.py --config 1 --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/bench_cfg1.json 2> gpurun_out/bench_cfg1.err; show gpurun_out/bench_cfg1.json cfg1
rm -rf gpurun_out/prof_kt
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_kt -o kt -- python bench.py --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/kt.log 2>&1
find gpurun_out/prof_kt -name '*kernel_stats.csv' -exec head -8 {} \;
