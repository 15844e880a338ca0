set -x
O=gpurun_out/r05g
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export SUSHI_BENCH_CACHE=/tmp/sbc
for D in 0 1 2 4 7; do
  SUSHI_HIP_TSPEC_DBG=$D timeout 150 python bench.py --steps 10 --warmup 3 --profile-only --exclusion band > $O/bench_dbg$D.json 2> $O/bench_dbg$D.err
  python - <<PY
import json
try:
    d=json.load(open("$O/bench_dbg$D.json")); r=d["roofline"]
    print("dbg$D", round(d["value"]), round(d["ms_per_step"],2), {k: round(v,2) for k,v in r["stage_ms"].items()}, r["diagnostics"]["pairs_transformed"])
except Exception as e: print("no line", e)
PY
done
OUT=r05kt2 bash tools/gpu_kt.sh 2>&1 | tail -12
