O=gpurun_out/r06j
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export SUSHI_BENCH_CACHE=/tmp/sushi_bench_cache
timeout 400 python bench.py --steps 3 --warmup 1 --profile-only --emulate-shards 0 > $O/warm.json 2> $O/warm.err
for rep in 1 2; do
timeout 200 python tools/stage_times.py --steps 10 --tag product | tee -a $O/la2.jsonl
SUSHI_HIP_LIB=$PWD/sushi_amd/lib/libsushi_hip_la2.so timeout 200 python tools/stage_times.py --steps 10 --tag la2 | tee -a $O/la2.jsonl
done
