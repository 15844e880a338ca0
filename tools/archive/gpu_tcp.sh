# TA / TCP counters of the default workload's kernels (three small passes; every profiler run under timeout)
set -x
mkdir -p gpurun_out/tcp
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export SUSHI_BENCH_CACHE=/tmp/sushi_bench_cache
timeout 150 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/tcp/base.json 2> gpurun_out/tcp/base.err
run() { n=$1; shift; rm -rf gpurun_out/prof_$n; timeout 55 rocprofv3 --pmc "$@" --output-format csv -d gpurun_out/prof_$n -o p -- python tools/stage_times.py --steps 2 --tag $n > gpurun_out/tcp/$n.log 2>&1; echo "pass $n rc=$?"; }
run ta TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_TOTAL_WAVEFRONTS_sum
run tcp1 TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCR_TCP_STALL_CYCLES_sum
run tcp2 TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum
python tools/summarize_pmc.py gpurun_out/tcp/summary.csv $(find gpurun_out/prof_ta gpurun_out/prof_tcp1 gpurun_out/prof_tcp2 -name '*counter_collection.csv')
grep -E "^kernel|mac_kernel|ifft_kernel" gpurun_out/tcp/summary.csv
