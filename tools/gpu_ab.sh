set -x
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in "" _nt "" _nt; do
SUSHI_HIP_LIB=$GRAFT_REPO_ROOT/sushi_amd/lib/libsushi_hip$v.so timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/bench_a.json 2>/dev/null
python -c "import json;d=json.load(open('gpurun_out/bench_a.json'));print('lib$v',d['value'],d['roofline']['stage_ms'])"
done
