# the configs[4] bench line (4-h 24 kHz streams, 5000 events) and the configs[2] line again (with roofline.traffic,
# once profiles/pmc_traffic.json carries the digest of the sources being run)
set -x
mkdir -p gpurun_out/r02
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export SUSHI_BENCH_CACHE=/tmp/sushi_bench_cache
O=gpurun_out/r02
timeout 1500 python bench.py --config 4 --steps 5 --warmup 2 > $O/bench_cfg4_n1.json 2> $O/bench_cfg4_n1.err; tail -3 $O/bench_cfg4_n1.err; head -c 1500 $O/bench_cfg4_n1.json
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_cfg2_n1.json 2> $O/bench_cfg2_n1.err; head -c 600 $O/bench_cfg2_n1.json
