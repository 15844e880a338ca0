set -x
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 -L > gpurun_out/counters_list.txt 2>&1
B="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --events 300 --ws-mb 4096"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_kt -o kt -- $B > gpurun_out/kt.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d gpurun_out/prof_pmc1 -o p1 -- $B > gpurun_out/p1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM --output-format csv -d gpurun_out/prof_pmc2 -o p2 -- $B > gpurun_out/p2.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof_pmc3 -o p3 -- $B > gpurun_out/p3.log 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d gpurun_out/prof_pmc4 -o p4 -- $B > gpurun_out/p4.log 2>&1
ls -R gpurun_out | head -50
