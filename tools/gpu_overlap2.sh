# mac (sub-batch n+1) beside ifft (sub-batch n) on two plain streams; ifft built with 4 / 3 / 2 workgroups per CU
set -x
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { SUSHI_HIP_LIB=$1 SUSHI_HIP_OVERLAP=$2 timeout 200 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --ws-mb $3 > gpurun_out/ov.json 2>gpurun_out/ov.err || tail -3 gpurun_out/ov.err
python -c "import json,sys;d=json.load(open('gpurun_out/ov.json'));print('OV', '$1', '$2', $3, round(d['value']), d['parity']['max_shift_err_samples_vs_planted'], d['ms_per_step'], d['roofline']['stage_ms'])" | tee -a gpurun_out/overlap2.txt; }
for ws in 2048 4096; do
run gpurun_scratch/libsushi_ov0.so 0 $ws
for lib in ov0 ov3 ov2; do run gpurun_scratch/libsushi_$lib.so -1 $ws; done
done
