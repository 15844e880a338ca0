set -x
O=gpurun_out/r05j
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export SUSHI_BENCH_CACHE=/tmp/sbc
run() { name=$1; shift
  timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline "$@" > $O/bench_$name.json 2> $O/bench_$name.err; tail -c 200 $O/bench_$name.err
  python - <<PY
import json
try:
    d=json.load(open("$O/bench_$name.json")); r=d["roofline"]; g=r["diagnostics"]
    print("$name", round(d["value"]), round(d["ms_per_step"],2), {k: round(v,2) for k,v in r["stage_ms"].items()}, {k: g[k] for k in ("pairs_transformed","excluded_audited","band","suspended","band_votes","max_slb_ratio_excluded","slb_violations","flagged")})
    print("   ", {k: d["parity"].get(k) for k in ("max_idx_err_vs_oracle_sample", "max_abs_score_err_vs_oracle_sample", "oracle_sample_searches","max_shift_err_samples_vs_planted")})
except Exception as e: print("$name no line", e)
PY
}
run auto
run snr6 --snr 6
run cfg4 --config 4 --steps 5
timeout 900 python -m pytest tests -m gpu -q --timeout 600 > $O/pytest_gpu.log 2>&1; tail -5 $O/pytest_gpu.log
