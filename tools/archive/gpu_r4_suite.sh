# Round 4: the whole GPU suite + a bench line on the product as it is (evidence for an adopted kernel change)
set -x
O=gpurun_out/${OUT:-r4suite}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export SUSHI_BENCH_CACHE=/tmp/sushi_bench_cache
timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1; tail -5 $O/pytest_gpu.log
timeout 400 python bench.py --steps 20 --warmup 5 ${BENCH_ARGS:---no-cpu-baseline --cpu-sample 256} > $O/bench_cfg2_n1.json 2> $O/b.err; tail -3 $O/b.err
python - <<PY
import json
d=json.load(open("$O/bench_cfg2_n1.json")); r=d["roofline"]; p=d["parity"]
print(round(d["value"]), round(d["ms_per_step"],2), r["stage_ms"], r["diagnostics"], p["oracle_sample_searches"], p.get("max_idx_err_vs_oracle_sample"), p.get("max_abs_score_err_vs_oracle_sample"))
PY
