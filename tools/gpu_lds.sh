# LDS / issue counters of the FFT path's kernels (dev)
set -x
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline"
rm -rf gpurun_out/prof_lds gpurun_out/prof_lds2
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --output-format csv -d gpurun_out/prof_lds -o lds -- $B > gpurun_out/lds.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC --output-format csv -d gpurun_out/prof_lds2 -o lds -- $B > gpurun_out/lds2.log 2>&1
python tools/summarize_pmc.py gpurun_out/lds_summary.csv gpurun_out/prof_lds/lds_counter_collection.csv gpurun_out/prof_lds2/lds_counter_collection.csv
grep "kernel,\|ifft\|mac_kernel" gpurun_out/lds_summary.csv
