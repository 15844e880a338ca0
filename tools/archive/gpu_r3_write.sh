# WRITE_SIZE in a pass of its own (the pass that also asked for TCC_HIT_sum / TCC_MISS_sum hung in rocprofv3's start-up this time
# and was cut by its timeout)
set -x
O=gpurun_out/r03final2
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export SUSHI_BENCH_CACHE=/tmp/sushi_bench_cache
B="python bench.py --steps 5 --warmup 1 --no-cpu-baseline --cpu-sample 2"
timeout 110 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/prof_write -o write -- $B > $O/write2.log 2>&1
python tools/summarize_pmc.py $O/cfg2_write_summary.csv $(find gpurun_out/prof_write -name '*counter_collection.csv'); cat $O/cfg2_write_summary.csv
