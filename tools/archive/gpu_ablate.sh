# mac_kernel ablations (variant libraries built with -DSUSHI_DEV_MAC_ABL=n; bit 1 = no Y traffic, 2 = no row traffic,
# 4 = no multiply-accumulates, 8 = nothing removed; every variant stops after the mac stage) timed by tools/stage_times.py
set -x
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export SUSHI_BENCH_CACHE=/tmp/sushi_bench_cache
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/abl_base.json 2> gpurun_out/abl_base.err
python -c "
import json;d=json.load(open('gpurun_out/abl_base.json'));print('product',round(d['ms_per_step'],2),{k:round(v,2) for k,v in d['roofline']['stage_ms'].items()})"
: > gpurun_out/ablate.jsonl
timeout 200 python tools/stage_times.py --tag product >> gpurun_out/ablate.jsonl 2>gpurun_out/abl.err
for v in ${VARIANTS:-8 1 2 3 4 7}; do
  SUSHI_HIP_LIB=$GRAFT_REPO_ROOT/sushi_amd/lib/libsushi_hip_abl$v.so timeout 200 python tools/stage_times.py --tag abl$v >> gpurun_out/ablate.jsonl 2>>gpurun_out/abl.err
done
cat gpurun_out/ablate.jsonl; tail -3 gpurun_out/abl.err
