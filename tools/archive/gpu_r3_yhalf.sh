# Experiment: Y as packed halves (tools/experiments/r03_y_half_width.patch, libsushi_hip_yhalf.so) against the product
set -x
O=gpurun_out/r03yh
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export SUSHI_BENCH_CACHE=/tmp/sushi_bench_cache
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --cpu-sample 256 --delta 1e-2 > $O/product_delta1e-2.json 2> $O/p.err; tail -1 $O/p.err
export SUSHI_HIP_LIB=$PWD/sushi_amd/lib/libsushi_hip_yhalf.so
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --cpu-sample 256 --delta 1e-2 > $O/yhalf_delta1e-2.json 2> $O/y.err; tail -3 $O/y.err
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --cpu-sample 256 --delta 2e-3 > $O/yhalf_delta2e-3.json 2> $O/y2.err; tail -3 $O/y2.err
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --cpu-sample 256 --delta 1e-2 --hard-frac 0.05 > $O/yhalf_hard_delta1e-2.json 2> $O/y3.err; tail -3 $O/y3.err
unset SUSHI_HIP_LIB
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --cpu-sample 256 --hard-frac 0.05 > $O/product_hard.json 2> $O/p3.err; tail -1 $O/p3.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r03yh/*.json")):
    try:
        d=json.load(open(f)); r=d["roofline"]; p=d["parity"]
        print(f.split("/")[-1], round(d["ms_per_step"],2), {k:round(v,2) for k,v in r["stage_ms"].items()}, r["diagnostics"], "oracle", p["oracle_sample_searches"], "idx_err", p.get("max_idx_err_vs_oracle_sample"), "score", p.get("max_score_err_over_tolerance_vs_oracle_sample"), "planted", p["max_shift_err_samples_vs_planted"])
    except Exception as e: print(f, "ERR", e)
PY
