# CU-masked overlap of mac (sub-batch n+1) with ifft (sub-batch n)
set -x
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { timeout 200 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --ws-mb $1 > gpurun_out/ov.json 2>gpurun_out/ov.err || tail -3 gpurun_out/ov.err
python -c "import json,sys;d=json.load(open('gpurun_out/ov.json'));print('OV', '$SUSHI_HIP_OVERLAP', '$SUSHI_HIP_OVERLAP_LAYOUT', $1, round(d['value']), d['parity'], d['roofline']['stage_ms'])" | tee -a gpurun_out/overlap.txt; }
export SUSHI_HIP_OVERLAP=0
run 2048; run 4096
for lay in 0 1; do
export SUSHI_HIP_OVERLAP_LAYOUT=$lay
for k in 32 64 96; do
export SUSHI_HIP_OVERLAP=$k
run 4096; run 8192
done
done
