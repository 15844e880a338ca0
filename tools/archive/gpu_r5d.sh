set -x
O=gpurun_out/r05d
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export SUSHI_BENCH_CACHE=/tmp/sbc
timeout 120 python tools/band_diag.py 2>&1 | grep -v amdgpu.ids | head -12 | tee $O/band_diag.txt
timeout 400 python -m pytest tests/test_half_front.py tests/test_pair_exclusion.py -m gpu -q > $O/pytest_excl.log 2>&1; tail -8 $O/pytest_excl.log
for M in band auto; do
  timeout 150 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --exclusion $M > $O/bench_$M.json 2> $O/bench_$M.err; tail -c 300 $O/bench_$M.err
  python - <<PY
import json
try:
    d=json.load(open("$O/bench_$M.json")); r=d["roofline"]
    print("$M", round(d["value"]), round(d["ms_per_step"],2), {k: round(v,2) for k,v in r["stage_ms"].items()}, r["diagnostics"])
    print({k: d["parity"][k] for k in ("max_idx_err_vs_oracle_sample", "max_abs_score_err_vs_oracle_sample", "oracle_sample_searches")})
except Exception as e: print("no line", e)
PY
done
