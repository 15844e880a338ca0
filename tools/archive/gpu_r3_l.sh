# scores of ifft_kernel through the LDS: stage times against libsushi_hip_prev.so first; the parity tests and a bench line only
# if the inverse transform gained at least 0.25 ms (GPU minutes are short)
set -x
O=gpurun_out/r3l
mkdir -p $O; rm -f $O/ab.log
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export SUSHI_BENCH_CACHE=/tmp/sushi_bench_cache
timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --cpu-sample 64 > $O/bench_cfg2_n1.json 2> $O/b.err; tail -3 $O/b.err
for v in prev product gq6 prev product gq6; do
  if [ $v = product ]; then unset SUSHI_HIP_LIB; else export SUSHI_HIP_LIB=$PWD/sushi_amd/lib/libsushi_hip_$v.so; fi
  timeout 120 python tools/stage_times.py --steps 10 --tag $v 2>/dev/null | tail -1 | tee -a $O/ab.log
done
unset SUSHI_HIP_LIB
GAIN=$(python - <<'PY'
import json
r=[json.loads(l) for l in open("gpurun_out/r3l/ab.log")]
m=lambda t: min(x["stage_ms"]["ifft"] for x in r if x["tag"]==t)
print(1 if m("prev")-m("product") >= 0.25 else 0)
PY
)
echo GAIN=$GAIN
if [ "$GAIN" = 1 ]; then
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_ccoeff.py tests/test_bound_stress.py -m gpu -q > $O/pytest_gpu.log 2>&1; tail -15 $O/pytest_gpu.log
fi
python - <<'PY'
import json
d=json.load(open("gpurun_out/r3l/bench_cfg2_n1.json")); r=d["roofline"]; p=d["parity"]
print(round(d["value"]), round(d["ms_per_step"],2), r["stage_ms"], r["diagnostics"], p["oracle_sample_searches"], p.get("max_idx_err_vs_oracle_sample"), p.get("max_abs_score_err_vs_oracle_sample"))
PY
