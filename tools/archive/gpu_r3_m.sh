# A/B of the product against libsushi_hip_prev.so (+ VARIANTS) on one box: stage times first; the parity tests and a bench line
# with a 256-search oracle sample only if the step gained at least GATE ms (GPU minutes are short)
set -x
O=gpurun_out/${OUT:-r3m}
mkdir -p $O; rm -f $O/ab.log
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export SUSHI_BENCH_CACHE=/tmp/sushi_bench_cache
timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --cpu-sample 64 > $O/bench_quick.json 2> $O/b.err; tail -3 $O/b.err
V="prev product $VARIANTS"
for v in $V $V; do
  if [ $v = product ]; then unset SUSHI_HIP_LIB; else export SUSHI_HIP_LIB=$PWD/sushi_amd/lib/libsushi_hip_$v.so; fi
  timeout 120 python tools/stage_times.py --steps 10 --tag $v 2>/dev/null | tail -1 | tee -a $O/ab.log
done
unset SUSHI_HIP_LIB
GAIN=$(python - <<PY
import json
r=[json.loads(l) for l in open("$O/ab.log")]
m=lambda t: min(sum(x["stage_ms"].values()) for x in r if x["tag"]==t)
print(1 if m("prev")-m("product") >= ${GATE:-0.3} else 0)
PY
)
echo GAIN=$GAIN
if [ "$GAIN" = 1 ]; then
timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --cpu-sample 256 > $O/bench_cfg2_n1.json 2> $O/b2.err; tail -3 $O/b2.err
timeout 900 python -m pytest ${TESTS:-tests/test_gpu_parity.py tests/test_ccoeff.py tests/test_bound_stress.py} -m gpu -q > $O/pytest_gpu.log 2>&1; tail -15 $O/pytest_gpu.log
fi
python - <<PY
import json,os
f="$O/bench_cfg2_n1.json" if os.path.exists("$O/bench_cfg2_n1.json") else "$O/bench_quick.json"
d=json.load(open(f)); r=d["roofline"]; p=d["parity"]
print(f, round(d["value"]), round(d["ms_per_step"],2), r["stage_ms"], r["diagnostics"], p["oracle_sample_searches"], p.get("max_idx_err_vs_oracle_sample"), p.get("max_abs_score_err_vs_oracle_sample"))
PY
