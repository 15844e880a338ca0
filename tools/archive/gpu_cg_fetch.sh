# L2-miss volume of mac_kernel under forced chunk_group values (variant library with the SUSHI_DEV_CG hook of
# tools/experiments/r02_mac_dev_hooks.patch).  FETCH_SIZE in a pass of its own; every profiler run under timeout.
set -x
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export SUSHI_BENCH_CACHE=/tmp/sushi_bench_cache
export SUSHI_HIP_LIB=$GRAFT_REPO_ROOT/sushi_amd/lib/libsushi_hip_cg.so
timeout 200 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/cgf_base.json 2> gpurun_out/cgf.err
for cg in 1 8 32; do
  rm -rf gpurun_out/prof_cg$cg
  SUSHI_DEV_CG=$cg timeout 100 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof_cg$cg -o f -- python tools/stage_times.py --steps 2 --tag cg$cg > gpurun_out/cgf_$cg.log 2>&1
  python tools/summarize_pmc.py gpurun_out/cgf_$cg.csv $(find gpurun_out/prof_cg$cg -name '*counter_collection.csv') > /dev/null; echo cg$cg; grep -E "mac_kernel" gpurun_out/cgf_$cg.csv; tail -1 gpurun_out/cgf_$cg.log | cut -c1-200
done
