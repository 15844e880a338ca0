# A/B on one box: VARIANTS (default "prev product") = libsushi_hip_<v>.so next to the product library, two rounds of stage
# times each; then a bench line with a 256-search oracle sample and the kernel-level parity tests of the product
set -x
O=gpurun_out/${OUT:-r3k}
mkdir -p $O; rm -f $O/ab.log
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export SUSHI_BENCH_CACHE=/tmp/sushi_bench_cache
timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --cpu-sample 256 > $O/bench_cfg2_n1.json 2> $O/b.err; tail -3 $O/b.err
V=${VARIANTS:-prev product}
for v in $V $V; do
  if [ $v = product ]; then unset SUSHI_HIP_LIB; else export SUSHI_HIP_LIB=$PWD/sushi_amd/lib/libsushi_hip_$v.so; fi
  timeout 120 python tools/stage_times.py --steps 10 --tag $v 2>/dev/null | tail -1 | tee -a $O/ab.log
done
unset SUSHI_HIP_LIB
if [ "$SKIP_TESTS" != 1 ]; then
timeout 900 python -m pytest ${TESTS:-tests/test_gpu_parity.py tests/test_ccoeff.py tests/test_bound_stress.py} -m gpu -q > $O/pytest_gpu.log 2>&1; tail -30 $O/pytest_gpu.log
fi
python - <<PY
import json
d=json.load(open("$O/bench_cfg2_n1.json")); r=d["roofline"]; p=d["parity"]
print(round(d["value"]), round(d["ms_per_step"],2), r["stage_ms"], r["diagnostics"], p["oracle_sample_searches"], p.get("max_idx_err_vs_oracle_sample"), p.get("max_abs_score_err_vs_oracle_sample"))
PY
