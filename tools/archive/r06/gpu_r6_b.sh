# Round 6: the whole GPU suite + the default bench line (tspec back at two workgroups per CU)
set -x
O=gpurun_out/r06b
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export SUSHI_BENCH_CACHE=/tmp/sushi_bench_cache
timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_cfg2.json 2> $O/bench_cfg2.err
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; tail -5 $O/pytest_gpu.log
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/bench_*.json")):
    try:
        d=json.load(open(f)); r=d["roofline"]; g=r.get("diagnostics") or {}
        print(f.split("/")[-1], round(d["value"]), round(d["ms_per_step"],3), {k: round(v,2) for k,v in (r.get("stage_ms") or {}).items()}, "pairs", g.get("pairs_transformed"), "idx_err", d["parity"].get("max_idx_err_vs_oracle_sample"))
    except Exception as e: print(f, "ERR", e)
PY
