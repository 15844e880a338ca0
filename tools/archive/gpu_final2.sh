# The secondary bench lines and the SQ instruction-count pass at the round's final sources (tools/gpu_final.sh did the
# primary ones).  Every command under its own timeout.
set -x
mkdir -p gpurun_out/r02f
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export SUSHI_BENCH_CACHE=/tmp/sushi_bench_cache
O=gpurun_out/r02f
B="python bench.py --steps 5 --warmup 1 --no-cpu-baseline"
timeout 200 $B > $O/warm.json 2> $O/warm.err
rm -rf gpurun_out/prof_sq2
timeout 150 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU --output-format csv -d gpurun_out/prof_sq2 -o sq2 -- $B > $O/sq2.log 2>&1
python tools/summarize_pmc.py $O/cfg2_sq_summary.csv $(find gpurun_out/prof_sq2 -name '*counter_collection.csv'); grep -E "kernel|mac|ifft" $O/cfg2_sq_summary.csv | cut -c1-300
timeout 200 python bench.py --config 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_cfg1_n1.json 2>/dev/null
timeout 200 python bench.py --sample-type uint8 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_cfg2_u8_n1.json 2>/dev/null
timeout 200 python bench.py --hard-frac 0.05 --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_cfg2_hard_n1.json 2>/dev/null
for f in cfg1 cfg2_u8 cfg2_hard; do python -c "
import json,sys;d=json.load(open(sys.argv[1]));r=d['roofline'];print(sys.argv[2],round(d['value']),round(d['ms_per_step'],2),{k:round(v,2) for k,v in r.get('stage_ms',{}).items()},round(r['frac'],3))" $O/bench_${f}_n1.json $f; done
