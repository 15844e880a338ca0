O=gpurun_out/r06k
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export SUSHI_BENCH_CACHE=/tmp/sushi_bench_cache
timeout 400 python bench.py --steps 3 --warmup 1 --profile-only --emulate-shards 0 > $O/warm.json 2> $O/warm.err
timeout 300 python tools/overlap_probe.py 2>&1 | grep -v amdgpu.ids | tee $O/overlap_probe.json
