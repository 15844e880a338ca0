# PMC passes (separate runs, --pmc only) on the default bench workload; summaries -> gpurun_out/pmc_*.csv
set -x
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export SUSHI_BENCH_CACHE=/tmp/sushi_bench_cache
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline ${BENCH_ARGS:-}"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof_fetch -o fetch -- $B > gpurun_out/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d gpurun_out/prof_write -o write -- $B > gpurun_out/write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d gpurun_out/prof_sq1 -o sq1 -- $B > gpurun_out/sq1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU --output-format csv -d gpurun_out/prof_sq2 -o sq2 -- $B > gpurun_out/sq2.log 2>&1
python tools/summarize_pmc.py gpurun_out/pmc_summary.csv $(find gpurun_out/prof_fetch gpurun_out/prof_write gpurun_out/prof_sq1 gpurun_out/prof_sq2 -name '*counter_collection.csv')
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/pmc_summary.csv')))
keys=[k for k in rows[0] if k not in('kernel','dispatches')]
for r in rows:
    if any(x in r['kernel'] for x in ('mac','ifft','tspec','refine')):
        print(r['kernel'], r['dispatches'], {k:r[k] for k in keys})
PY
