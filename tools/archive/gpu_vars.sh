# A/B of variant libraries on the default bench line: VARIANTS="a b" -> sushi_amd/lib/libsushi_hip_<v>.so
set -x
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export SUSHI_BENCH_CACHE=/tmp/sushi_bench_cache
show() { python -c "
import json,sys;d=json.load(open(sys.argv[1]));r=d['roofline'];print(sys.argv[2],round(d['value']),round(d['ms_per_step'],2),{k:round(v,2) for k,v in r['stage_ms'].items()})" $1 $2; }
timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/v_base.json 2> gpurun_out/v.err; show gpurun_out/v_base.json product
for v in ${VARIANTS:-}; do
  SUSHI_HIP_LIB=$GRAFT_REPO_ROOT/sushi_amd/lib/libsushi_hip_$v.so timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/v_$v.json 2>> gpurun_out/v.err; show gpurun_out/v_$v.json $v
done
timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/v_base2.json 2>> gpurun_out/v.err; show gpurun_out/v_base2.json product_again
