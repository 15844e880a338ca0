# workspace-size sweep: does the Y round trip stay in the 256 MB memory-side cache when sub-batches are small?
set -x
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for lib in "" gpurun_scratch/libsushi_plain.so; do
for ws in 96 160 224 320 512 2048; do
SUSHI_HIP_LIB=$lib timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --ws-mb $ws > gpurun_out/ws.json 2>/dev/null
python -c "import json,sys;d=json.load(open('gpurun_out/ws.json'));print('WS', '$lib', $ws, round(d['value']), d['roofline']['stage_ms'])" | tee -a gpurun_out/ws_sweep.txt
done
done
