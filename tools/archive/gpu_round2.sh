# Round-2 evidence run on the MI355X box: GPU tests, smoke, bench (north-star configuration + the other workloads),
# rocprofv3 kernel trace + PMC passes.  Everything lands in gpurun_out/ (copied into profiles/r02/ afterwards).
set -x
mkdir -p gpurun_out/r02
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export SUSHI_BENCH_CACHE=/tmp/sushi_bench_cache
O=gpurun_out/r02
timeout 1200 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
show() { python -c "
import json,sys;d=json.load(open(sys.argv[1]));r=d['roofline'];print(sys.argv[2],round(d['value']),round(d['ms_per_step'],2),{k:round(v,2) for k,v in r.get('stage_ms',{}).items()},round(r['frac'],3),r.get('diagnostics'),d['parity'].get('max_idx_err_vs_oracle_sample'))" $1 $2; }
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_cfg2_n1.json 2> $O/bench_cfg2_n1.err; show $O/bench_cfg2_n1.json cfg2; tail -2 $O/bench_cfg2_n1.err
timeout 600 python bench.py --config 1 --steps 20 --warmup 5 > $O/bench_cfg1_n1.json 2>/dev/null; show $O/bench_cfg1_n1.json cfg1
timeout 600 python bench.py --sample-type uint8 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_cfg2_u8_n1.json 2>/dev/null; show $O/bench_cfg2_u8_n1.json cfg2_u8
timeout 600 python bench.py --hard-frac 0.05 --steps 10 --warmup 2 > $O/bench_cfg2_hard_n1.json 2>/dev/null; show $O/bench_cfg2_hard_n1.json cfg2_hard
timeout 600 python bench.py --path direct --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_cfg2_direct_n1.json 2>/dev/null; show $O/bench_cfg2_direct_n1.json cfg2_direct
if [ "${WITH_CFG4:-0}" = "1" ]; then
  timeout 900 python bench.py --config 4 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_cfg4_n1.json 2> $O/bench_cfg4_n1.err; show $O/bench_cfg4_n1.json cfg4; tail -2 $O/bench_cfg4_n1.err
fi
B="python bench.py --steps 5 --warmup 1 --no-cpu-baseline"
rm -rf gpurun_out/prof_*
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_kt -o kt -- $B > $O/kt.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof_fetch -o fetch -- $B > $O/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d gpurun_out/prof_write -o write -- $B > $O/write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d gpurun_out/prof_sq1 -o sq1 -- $B > $O/sq1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU --output-format csv -d gpurun_out/prof_sq2 -o sq2 -- $B > $O/sq2.log 2>&1
cp gpurun_out/prof_kt/kt_kernel_stats.csv $O/cfg2_kernel_stats.csv; cp gpurun_out/prof_kt/kt_domain_stats.csv $O/cfg2_domain_stats.csv
python tools/summarize_pmc.py $O/cfg2_pmc_summary.csv $(find gpurun_out/prof_fetch gpurun_out/prof_write gpurun_out/prof_sq1 gpurun_out/prof_sq2 -name '*counter_collection.csv')
head -8 $O/cfg2_kernel_stats.csv; grep -E "kernel|mac|ifft" $O/cfg2_pmc_summary.csv
rm -rf gpurun_out/prof_hard; rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_hard -o hard -- python bench.py --hard-frac 0.05 --steps 3 --warmup 1 --no-cpu-baseline > $O/hard_kt.log 2>&1; cp gpurun_out/prof_hard/hard_kernel_stats.csv $O/cfg2_hard_kernel_stats.csv; head -9 $O/cfg2_hard_kernel_stats.csv
timeout 300 python tools/latency.py > $O/latency.json 2> $O/latency.err; tail -c 600 $O/latency.json
