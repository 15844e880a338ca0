# Round 4: SQ instruction / wait counters of mac_kernel, product against a variant library (V)
set -x
O=gpurun_out/${OUT:-r4sq}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export SUSHI_BENCH_CACHE=/tmp/sushi_bench_cache
timeout 60 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --cpu-sample 16 > $O/bench.json 2> $O/b.err   # (writes the stream cache)
for v in product $V; do
  if [ $v = product ]; then unset SUSHI_HIP_LIB; else export SUSHI_HIP_LIB=$PWD/sushi_amd/lib/libsushi_hip_$v.so; fi
  timeout 70 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VALU SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU --kernel-include-regex "mac_kernel" --output-format csv -d gpurun_out/prof_sq_$v -o sq -- python tools/stage_times.py --steps 1 --tag sq_$v > $O/sq_$v.log 2>&1
  python tools/summarize_pmc.py $O/sq_$v.csv $(find gpurun_out/prof_sq_$v -name '*counter_collection.csv'); cat $O/sq_$v.csv
done
