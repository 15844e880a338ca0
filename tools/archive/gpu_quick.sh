# quick check after a kernel change: the parity tests, then bench.py on the product library and on variant libraries
set -x
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export SUSHI_BENCH_CACHE=/tmp/sushi_bench_cache
timeout 900 python -m pytest ${TESTS:-tests/test_gpu_parity.py} -m gpu -q -x > gpurun_out/pytest_quick.log 2>&1; tail -4 gpurun_out/pytest_quick.log
show() { python -c "
import json,sys;d=json.load(open(sys.argv[1]));r=d['roofline'];print(sys.argv[2],round(d['value']),round(d['ms_per_step'],2),{k:round(v,2) for k,v in r['stage_ms'].items()},r.get('diagnostics',{}).get('flagged'))" $1 $2; }
timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline ${BENCH_ARGS:-} > gpurun_out/q_base.json 2>gpurun_out/q.err; show gpurun_out/q_base.json base
for v in ${VARIANTS:-}; do
  SUSHI_HIP_LIB=$GRAFT_REPO_ROOT/sushi_amd/lib/libsushi_hip_$v.so timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline ${BENCH_ARGS:-} > gpurun_out/q_$v.json 2>>gpurun_out/q.err; show gpurun_out/q_$v.json $v
done
tail -3 gpurun_out/q.err
