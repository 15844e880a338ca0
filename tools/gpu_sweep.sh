# sweep-aligned mac starts: A/B at configs[1] and configs[2] sizes, then the FFT parity tests
set -x
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for lib in gpurun_scratch/libsushi_nosweep.so sushi_amd/lib/libsushi_hip.so; do
SUSHI_HIP_LIB=$lib timeout 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline > gpurun_out/sw.json 2>/dev/null
python -c "import json;d=json.load(open('gpurun_out/sw.json'));print('SW cfg1', '$lib', round(d['value']), d['roofline']['stage_ms']['mac'], d['roofline']['stage_ms']['ifft'])"
SUSHI_HIP_LIB=$lib timeout 300 python bench.py --window 120 --minutes 120 --events 375 --steps 8 --warmup 3 --no-cpu-baseline > gpurun_out/sw.json 2>/dev/null
python -c "import json;d=json.load(open('gpurun_out/sw.json'));print('SW cfg3', '$lib', round(d['value']), d['roofline']['stage_ms']['mac'], d['roofline']['stage_ms']['ifft'])"
done
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q 2>&1 | tail -2
