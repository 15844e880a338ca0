# kernel trace of the default workload (per-kernel averages)
set -x
O=gpurun_out/${OUT:-r06kt}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export SUSHI_BENCH_CACHE=/tmp/sushi_bench_cache
B="python bench.py --steps 5 --warmup 1 --profile-only --emulate-shards 0 $ARGS"
timeout 400 $B > $O/warm.json 2> $O/warm.err
rm -rf gpurun_out/prof_kt
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_kt -o kt -- $B > $O/kt.log 2>&1
cp gpurun_out/prof_kt/kt_kernel_stats.csv $O/kernel_stats.csv
python - <<PY
import csv
rows=list(csv.DictReader(open("$O/kernel_stats.csv")))
for r in rows[:28]:
    print("%-60s calls %5s avg_us %10.1f total_ms %9.3f" % (r["Name"][:60], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e6))
PY
