# Round 4: bench.py itself (parity sample + stage times) under variant libraries VS="a b ..."; K (optional): a few kernel-level tests under the first
set -x
O=gpurun_out/${OUT:-r4vb}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export SUSHI_BENCH_CACHE=/tmp/sushi_bench_cache
for V in $VS; do
  export SUSHI_HIP_LIB=$PWD/sushi_amd/lib/libsushi_hip_$V.so
  timeout 60 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --cpu-sample ${SAMPLE:-64} > $O/bench_$V.json 2> $O/b_$V.err; tail -n 2 $O/b_$V.err
  if [ -n "$K" ]; then timeout 100 python -m pytest tests/test_gpu_parity.py tests/test_pair_exclusion.py -m gpu -q -x -k "$K" > $O/pytest_$V.log 2>&1; tail -n 4 $O/pytest_$V.log; K=; fi
done
python - <<PY
import json
for V in "$VS".split():
    try:
        d=json.load(open("$O/bench_%s.json" % V)); r=d["roofline"]; p=d["parity"]
        print(V, round(d["value"]), round(d["ms_per_step"],2), {k: round(v,3) for k,v in r["stage_ms"].items()}, r["diagnostics"]["flagged"], p["oracle_sample_searches"], p.get("max_idx_err_vs_oracle_sample"))
    except Exception as e: print(V, "ERR", e)
PY
