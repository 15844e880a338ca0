# dev: mac-stage-only variant libraries (SUSHI_DEV_STOP_AFTER_MAC), optionally with SUSHI_HIP_MAC_CG sweeps
set -x
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export SUSHI_BENCH_CACHE=/tmp/sushi_bench_cache
show() { python -c "
import json,sys;d=json.load(open(sys.argv[1]));r=d['roofline'];print(sys.argv[2],round(d['ms_per_step'],2),{k:round(v,2) for k,v in r['stage_ms'].items()})" $1 $2; }
for v in ${VARIANTS:-}; do
  SUSHI_HIP_LIB=$GRAFT_REPO_ROOT/sushi_amd/lib/libsushi_hip_$v.so timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --skip-verify ${BENCH_ARGS:-} > gpurun_out/md_$v.json 2>>gpurun_out/md.err; show gpurun_out/md_$v.json $v
done
for cg in ${CGS:-}; do
  SUSHI_HIP_MAC_CG=$cg SUSHI_HIP_LIB=$GRAFT_REPO_ROOT/sushi_amd/lib/libsushi_hip_m0.so timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --skip-verify ${BENCH_ARGS:-} > gpurun_out/md_cg$cg.json 2>>gpurun_out/md.err; show gpurun_out/md_cg$cg.json cg$cg
done
tail -3 gpurun_out/md.err
