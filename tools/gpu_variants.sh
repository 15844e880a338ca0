# bench.py over variant libraries (VARIANTS="abl1 abl2 ..."; each sushi_amd/lib/libsushi_hip_<v>.so), dev A/B
set -x
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export SUSHI_BENCH_CACHE=/tmp/sushi_bench_cache
show() { python -c "
import json,sys;d=json.load(open(sys.argv[1]));r=d['roofline'];print(sys.argv[2],round(d['value']),round(d['ms_per_step'],2),{k:round(v,2) for k,v in r['stage_ms'].items()})" $1 $2; }
python bench.py --steps 10 --warmup 2 --no-cpu-baseline ${BENCH_ARGS:-} > gpurun_out/var_base.json 2>gpurun_out/var.err; show gpurun_out/var_base.json base
for v in ${VARIANTS:-}; do
  SUSHI_HIP_LIB=$GRAFT_REPO_ROOT/sushi_amd/lib/libsushi_hip_$v.so python bench.py --steps 10 --warmup 2 --no-cpu-baseline --skip-verify ${BENCH_ARGS:-} > gpurun_out/var_$v.json 2>>gpurun_out/var.err; show gpurun_out/var_$v.json $v
done
tail -3 gpurun_out/var.err
