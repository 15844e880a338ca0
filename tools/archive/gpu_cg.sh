set -x
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export SUSHI_BENCH_CACHE=/tmp/sushi_bench_cache
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/cg_base.json 2> gpurun_out/cg.err
: > gpurun_out/cg.jsonl
for cg in 1 2 4 8 16 32; do
  SUSHI_DEV_CG=$cg timeout 200 python tools/stage_times.py --tag cg$cg >> gpurun_out/cg.jsonl 2>>gpurun_out/cg.err
done
cat gpurun_out/cg.jsonl
