# lanes inside the library: tests first, then the bench job under a sweep of SUSHI_HIP_LANES (subs:lanes)
O=gpurun_out/r06l
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export SUSHI_BENCH_CACHE=/tmp/sushi_bench_cache
timeout 900 python -m pytest tests/test_pair_exclusion.py tests/test_native_abi.py -m gpu -x -q -k "lanes or reset or abi" > $O/pytest_lanes.log 2>&1; tail -5 $O/pytest_lanes.log
timeout 400 python bench.py --steps 3 --warmup 1 --profile-only --emulate-shards 0 > $O/warm.json 2> $O/warm.err
for rep in 1 2; do
for L in 1:1 auto 6:3 3:3 4:2 9:3 12:3 8:4 12:4 6:2 2:2; do
  if [ $L = auto ]; then unset SUSHI_HIP_LANES; else export SUSHI_HIP_LANES=$L; fi
  timeout 200 python bench.py --steps 20 --warmup 3 --profile-only --emulate-shards 0 > $O/b.json 2> $O/b.err
  python - <<PY | tee -a $O/sweep.txt
import json
d=json.load(open("$O/b.json")); r=d["roofline"]
print("$L", round(d["ms_per_step"],3), {k: round(v,2) for k,v in (r.get("stage_ms") or {}).items()}, (r.get("diagnostics") or {}).get("pairs_transformed"), d["parity"].get("planted_ok", d["parity"]))
PY
done
done
unset SUSHI_HIP_LANES
