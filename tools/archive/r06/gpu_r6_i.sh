# Round 6: bound kernels without the row energies in the worst-case mode -- suite subset + lines that use each form
set -x
O=gpurun_out/r06i
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export SUSHI_BENCH_CACHE=/tmp/sushi_bench_cache
timeout 900 python -m pytest tests/test_pair_exclusion.py tests/test_bound_stress.py tests/test_ccoeff.py -m gpu -q -x > $O/pytest_sel.log 2>&1; tail -3 $O/pytest_sel.log
for wl in "cfg2:" "whole:--exclusion whole" "snr0:--snr 0" "stat20:" ; do
  name=${wl%%:*}; args=${wl#*:}
  case "$name" in stat*) export SUSHI_HIP_BOUND_MODEL=statistical;; *) unset SUSHI_HIP_BOUND_MODEL;; esac
  timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --emulate-shards 0 $args > $O/bench_$name.json 2> $O/bench_$name.err || tail -3 $O/bench_$name.err
done
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/bench_*.json")):
    try:
        d=json.load(open(f)); r=d["roofline"]; g=r.get("diagnostics") or {}; p=d["parity"]
        print(f.split("/")[-1], round(d["value"]), round(d["ms_per_step"],3), {k: round(v,2) for k,v in (r.get("stage_ms") or {}).items()}, "pairs", g.get("pairs_transformed"), "band", g.get("band"), "idx_err", p.get("max_idx_err_vs_oracle_sample"))
    except Exception as e: print(f, "ERR", e)
PY
