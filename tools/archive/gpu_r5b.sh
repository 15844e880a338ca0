# round 5, second call: the band-split exclusion's first run on the GPU
set -x
O=gpurun_out/r05b
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export SUSHI_BENCH_CACHE=/tmp/sbc
timeout 60 tools/ubench/half_front_check > $O/half_front.log 2>&1; tail -3 $O/half_front.log
timeout 300 python -m pytest tests/test_half_front.py tests/test_pair_exclusion.py -m gpu -q -x > $O/pytest_excl.log 2>&1; tail -15 $O/pytest_excl.log
for M in band whole; do
  timeout 120 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --exclusion $M > $O/bench_$M.json 2> $O/bench_$M.err; tail -c 300 $O/bench_$M.err
  python - <<PY
import json
try:
    d=json.load(open("$O/bench_$M.json")); r=d["roofline"]
    print("$M", round(d["value"]), round(d["ms_per_step"],2), {k: round(v,2) for k,v in r["stage_ms"].items()}, r["diagnostics"])
    print(d["parity"])
except Exception as e: print("no line", e)
PY
done
timeout 120 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_auto.json 2> $O/bench_auto.err; tail -c 300 $O/bench_auto.err
python - <<PY
import json
try:
    d=json.load(open("$O/bench_auto.json")); r=d["roofline"]
    print("auto", round(d["value"]), round(d["ms_per_step"],2), {k: round(v,2) for k,v in r["stage_ms"].items()}, r["diagnostics"])
except Exception as e: print("no line", e)
PY
