# Round-1 evidence run on the MI355X box: GPU tests, smoke, bench (both paths), rocprofv3 kernel trace + PMC passes.
set -x
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu.log 2>&1; tail -3 gpurun_out/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 600 python bench.py --steps 20 --warmup 3 > gpurun_out/bench_fft_n1.json 2> gpurun_out/bench_fft_n1.err; cat gpurun_out/bench_fft_n1.json
timeout 600 python bench.py --path direct --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench_direct_n1.json 2>/dev/null; cut -c1-300 gpurun_out/bench_direct_n1.json
timeout 600 python bench.py --sample-type uint8 --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/bench_fft_u8_n1.json 2>/dev/null; python -c "import json;d=json.load(open('gpurun_out/bench_fft_u8_n1.json'));print('u8',d['value'],d['roofline']['stage_ms'],d['parity'])"
timeout 600 python bench.py --window 120 --minutes 120 --events 375 --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/bench_fft_cfg3_n1.json 2>/dev/null; python -c "import json;d=json.load(open('gpurun_out/bench_fft_cfg3_n1.json'));print('cfg3',d['value'],d['roofline']['stage_ms'],d['parity'])"
B="python bench.py --steps 5 --warmup 1 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_kt -o kt -- $B > gpurun_out/kt.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof_fetch -o fetch -- $B > gpurun_out/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d gpurun_out/prof_write -o write -- $B > gpurun_out/write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d gpurun_out/prof_sq1 -o sq1 -- $B > gpurun_out/sq1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU --output-format csv -d gpurun_out/prof_sq2 -o sq2 -- $B > gpurun_out/sq2.log 2>&1
