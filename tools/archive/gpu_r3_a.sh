# Round 3, first GPU call: cv2 probe, the whole GPU suite (new full-size uint8 / hard-material tests), the default bench line
# with the reworked oracle leg, SQ counter passes that attribute ifft_kernel's / mac_kernel's time, the VALU issue-rate ubench.
set -x
O=gpurun_out/r03a
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export SUSHI_BENCH_CACHE=/tmp/sushi_bench_cache
python -c "import cv2; print('cv2', cv2.__version__)" > $O/cv2_probe.log 2>&1; cat $O/cv2_probe.log | tail -1
nproc > $O/nproc.log; cat $O/nproc.log
timeout 420 python bench.py --steps 20 --warmup 5 > $O/bench_cfg2_n1.json 2> $O/bench_cfg2_n1.err; head -c 600 $O/bench_cfg2_n1.json; echo; tail -3 $O/bench_cfg2_n1.err
timeout 60 tools/ubench/valu_rate > $O/valu_rate.log 2>&1; cat $O/valu_rate.log
run() { n=$1; shift; rm -rf gpurun_out/prof_$n; timeout 100 rocprofv3 --pmc "$@" --kernel-include-regex "ifft_kernel|mac_kernel" --output-format csv -d gpurun_out/prof_$n -o p -- python tools/stage_times.py --steps 2 --tag $n > $O/$n.log 2>&1; echo "pass $n rc=$?"; tail -1 $O/$n.log; }
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY
run sq2 SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS
run sq3 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE
run sq4 SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT
run grbm GRBM_GUI_ACTIVE GRBM_COUNT
python tools/summarize_pmc.py $O/cfg2_sq_summary.csv $(find gpurun_out/prof_sq1 gpurun_out/prof_sq2 gpurun_out/prof_sq3 gpurun_out/prof_sq4 gpurun_out/prof_grbm -name '*counter_collection.csv')
cat $O/cfg2_sq_summary.csv
timeout 900 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1; tail -5 $O/pytest_gpu.log
