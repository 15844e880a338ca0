set -x
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export SUSHI_BENCH_CACHE=/tmp/sushi_bench_cache
timeout 1200 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu.log 2>&1; tail -5 gpurun_out/pytest_gpu.log
show() { python -c "
import json,sys;d=json.load(open(sys.argv[1]));r=d['roofline'];print(sys.argv[2],round(d['value']),round(d['ms_per_step'],2),{k:round(v,2) for k,v in r['stage_ms'].items()},round(r['frac'],3),d['parity'])" $1 $2; }
timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/l_cfg2.json 2> gpurun_out/l_cfg2.err; show gpurun_out/l_cfg2.json cfg2; tail -2 gpurun_out/l_cfg2.err
timeout 1500 python bench.py --config 4 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/l_cfg4.json 2> gpurun_out/l_cfg4.err; show gpurun_out/l_cfg4.json cfg4; tail -2 gpurun_out/l_cfg4.err
