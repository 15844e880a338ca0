set -x
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export SUSHI_BENCH_CACHE=/tmp/sushi_bench_cache
show() { python -c "
import json,sys;d=json.load(open(sys.argv[1]));r=d['roofline'];print(sys.argv[2],round(d['value']),round(d['ms_per_step'],2),{k:round(v,2) for k,v in r['stage_ms'].items()})" $1 $2; }
python bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/sw_auto.json 2>gpurun_out/sw.err; show gpurun_out/sw_auto.json auto
for cg in 1 2 4 8 16; do
  SUSHI_HIP_MAC_CG=$cg python bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/sw_$cg.json 2>>gpurun_out/sw.err; show gpurun_out/sw_$cg.json cg$cg
done
for cg in 2 4 8; do
  SUSHI_HIP_MAC_CG=$cg python bench.py --config 1 --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/sw1_$cg.json 2>>gpurun_out/sw.err; show gpurun_out/sw1_$cg.json cfg1_cg$cg
done
python bench.py --config 1 --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/sw1_auto.json 2>>gpurun_out/sw.err; show gpurun_out/sw1_auto.json cfg1_auto
tail -3 gpurun_out/sw.err
