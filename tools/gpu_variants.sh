# A/B of library variants built under gpurun_scratch/ (dev): bench each, first against the parity sample
set -x
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for lib in $LIBS; do
for rep in 1 2; do
SUSHI_HIP_LIB=$lib timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/var.json 2>gpurun_out/var.err || tail -3 gpurun_out/var.err
python -c "import json,sys;d=json.load(open('gpurun_out/var.json'));print('VAR', '$lib', round(d['value']), d['parity']['max_shift_err_samples_vs_planted'], d['roofline']['stage_ms'])" | tee -a gpurun_out/variants.txt
done
done
