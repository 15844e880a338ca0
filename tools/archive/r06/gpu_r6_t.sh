# dense searches regrouped into items of their own: tests, then the lines it is for
O=gpurun_out/r06t
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export SUSHI_BENCH_CACHE=/tmp/sushi_bench_cache
timeout 900 python -m pytest tests/test_pair_exclusion.py tests/test_bound_stress.py -m gpu -x -q > $O/pytest.log 2>&1; tail -5 $O/pytest.log
line() {   # name args...
  name=$1; shift 1
  timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --cpu-sample 128 --emulate-shards 0 "$@" > $O/b_$name.json 2> $O/b.err
  python - <<PY | tee -a $O/sweep.txt
import json
try:
    d=json.load(open("$O/b_$name.json")); r=d["roofline"]; g=r.get("diagnostics") or {}; p=d["parity"]
    print("$name", round(d["ms_per_step"],3), {k: round(v,2) for k,v in (r.get("stage_ms") or {}).items()}, g.get("pairs_transformed"), "oracle", p["oracle_sample_searches"], p.get("max_idx_err_vs_oracle_sample"))
except Exception as e:
    print("$name", "FAILED", e, open("$O/b.err").read()[-600:])
PY
}
line cfg2
line dub --source dub
line cfg2
SUSHI_HIP_LANES=1:1 line cfg2_lanes1
line partial --source partial
line snr6 --snr 6
line dub_cc --source dub --method ccoeff_normed
