# Round 4: kernel-level parity tests of the product + stage times + optional extra bench lines (EXTRA="--method ccoeff_normed|--hard-frac 0.05|...")
set -x
O=gpurun_out/${OUT:-r4q}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export SUSHI_BENCH_CACHE=/tmp/sushi_bench_cache
timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --cpu-sample ${SAMPLE:-256} > $O/bench_cfg2_n1.json 2> $O/b.err; tail -3 $O/b.err
if [ "$SKIP_TESTS" != 1 ]; then
timeout 900 python -m pytest ${TESTS:-tests/test_gpu_parity.py tests/test_ccoeff.py tests/test_bound_stress.py} -m gpu -q -x > $O/pytest_gpu.log 2>&1; tail -5 $O/pytest_gpu.log
fi
IFS='|' read -ra EX <<< "$EXTRA"
k=0
for e in "${EX[@]}"; do
  k=$((k+1))
  timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --cpu-sample 64 $e > $O/bench_extra$k.json 2> $O/e$k.err; tail -2 $O/e$k.err
done
for v in $VARIANTS; do
  if [ $v = product ]; then unset SUSHI_HIP_LIB; else export SUSHI_HIP_LIB=$PWD/sushi_amd/lib/libsushi_hip_$v.so; fi
  timeout 120 python tools/stage_times.py --steps 10 --tag $v 2>/dev/null | tail -1 | tee -a $O/ab.log
done
unset SUSHI_HIP_LIB
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/bench_*.json")):
    try:
        d=json.load(open(f)); r=d["roofline"]; p=d["parity"]
        print(f.split("/")[-1], d["config"].get("method"), round(d["value"]), round(d["ms_per_step"],2), {k: round(v,3) for k,v in r["stage_ms"].items()}, r["diagnostics"], p["oracle_sample_searches"], p.get("max_idx_err_vs_oracle_sample"), p.get("max_abs_score_err_vs_oracle_sample"))
    except Exception as e: print(f, "ERR", e)
PY
