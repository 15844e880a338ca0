# Round 4: rocprofv3 kernel traces (per-kernel time) of bench workloads.  WL="name:args|name:args"
set -x
O=gpurun_out/${OUT:-r4trace}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export SUSHI_BENCH_CACHE=/tmp/sushi_bench_cache
IFS='|' read -ra W <<< "$WL"
for w in "${W[@]}"; do
  name=${w%%:*}; args=${w#*:}
  B="python bench.py --steps 5 --warmup 1 --profile-only $args"
  timeout 300 $B > $O/warm_$name.json 2> $O/warm_$name.err
  rm -rf gpurun_out/prof_kt
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_kt -o kt -- $B > $O/kt_$name.log 2>&1
  cp gpurun_out/prof_kt/kt_kernel_stats.csv $O/${name}_kernel_stats.csv
  head -14 $O/${name}_kernel_stats.csv | cut -c1-160
done
