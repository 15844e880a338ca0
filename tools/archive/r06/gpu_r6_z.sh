# do the lanes share hardware queues?  GPU_MAX_HW_QUEUES (read by the HIP runtime at start-up) against the step, and the timeline's concurrency
O=gpurun_out/r06z
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export SUSHI_BENCH_CACHE=/tmp/sushi_bench_cache
timeout 400 python bench.py --steps 3 --warmup 1 --profile-only --emulate-shards 0 > /dev/null 2>&1
for rep in 1 2; do
for Q in default 2 4 8 16; do
  if [ $Q = default ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$Q; fi
  for L in auto 12:4 1:1; do
  if [ $L = auto ]; then unset SUSHI_HIP_LANES; else export SUSHI_HIP_LANES=$L; fi
  timeout 200 python bench.py --steps 20 --warmup 3 --profile-only --emulate-shards 0 > $O/b.json 2> $O/b.err
  python -c "
import json; d=json.load(open('$O/b.json')); print('queues=$Q lanes=$L', round(d['ms_per_step'],3))" | tee -a $O/sweep.txt
  done
done
done
unset SUSHI_HIP_LANES
for Q in default 8; do
  if [ $Q = default ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$Q; fi
  rm -rf gpurun_out/prof_tl
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_tl -o kt -- python bench.py --steps 5 --warmup 1 --profile-only --emulate-shards 0 > /dev/null 2>&1
  echo "queues=$Q"; python tools/timeline.py $(find gpurun_out/prof_tl -name "*kernel_trace.csv" | head -1) | tail -1
done
