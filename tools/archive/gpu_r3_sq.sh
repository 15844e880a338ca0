# SQ counters of ifft_kernel / mac_kernel after the wave plan + packed-half Y (compare profiles/r03/cfg2_sq_summary_before_waveplan.csv)
set -x
O=gpurun_out/r03sq
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export SUSHI_BENCH_CACHE=/tmp/sushi_bench_cache
timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --cpu-sample 2 > $O/warm.json 2> $O/warm.err
run() { n=$1; shift; rm -rf gpurun_out/prof_$n; timeout 100 rocprofv3 --pmc "$@" --kernel-include-regex "ifft_kernel|mac_kernel" --output-format csv -d gpurun_out/prof_$n -o p -- python tools/stage_times.py --steps 2 --tag $n > $O/$n.log 2>&1; echo "pass $n rc=$?"; }
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY
run sq2 SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS
run sq3 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE
run sq4 SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT
run grbm GRBM_GUI_ACTIVE GRBM_COUNT
python tools/summarize_pmc.py $O/cfg2_sq_summary.csv $(find gpurun_out/prof_sq1 gpurun_out/prof_sq2 gpurun_out/prof_sq3 gpurun_out/prof_sq4 gpurun_out/prof_grbm -name '*counter_collection.csv')
cat $O/cfg2_sq_summary.csv
