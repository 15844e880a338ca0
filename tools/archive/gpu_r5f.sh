set -x
O=gpurun_out/r05f
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export SUSHI_BENCH_CACHE=/tmp/sbc
timeout 150 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_auto.json 2> $O/bench_auto.err; tail -c 300 $O/bench_auto.err
python - <<PY
import json
try:
    d=json.load(open("$O/bench_auto.json")); r=d["roofline"]
    print("auto", round(d["value"]), round(d["ms_per_step"],2), {k: round(v,2) for k,v in r["stage_ms"].items()}, r["diagnostics"])
    print({k: d["parity"][k] for k in ("max_idx_err_vs_oracle_sample", "max_abs_score_err_vs_oracle_sample", "oracle_sample_searches")})
except Exception as e: print("no line", e)
PY
timeout 800 python -m pytest tests -m gpu -q --timeout 600 > $O/pytest_gpu.log 2>&1; tail -8 $O/pytest_gpu.log
