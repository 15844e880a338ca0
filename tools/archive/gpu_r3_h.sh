set -x
O=gpurun_out/r03h
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export SUSHI_BENCH_CACHE=/tmp/sushi_bench_cache
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --cpu-sample 512 > $O/bench_cfg2_n1.json 2> $O/b.err; tail -2 $O/b.err
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_ccoeff.py tests/test_shifts.py tests/test_shifts_golden.py -m gpu -q -x > $O/pytest_gpu.log 2>&1; tail -6 $O/pytest_gpu.log
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --cpu-sample 256 --sample-type uint8 > $O/bench_cfg2_u8_n1.json 2> $O/b2.err; tail -2 $O/b2.err
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --cpu-sample 64 --hard-frac 0.05 > $O/bench_cfg2_hard_n1.json 2> $O/b3.err; tail -2 $O/b3.err
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --cpu-sample 256 --method ccoeff_normed > $O/bench_cfg2_ccoeff_n1.json 2> $O/b4.err; tail -2 $O/b4.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r03h/*.json")):
    try:
        d=json.load(open(f)); r=d["roofline"]; p=d["parity"]
        print(f.split("/")[-1], round(d["value"]), round(d["ms_per_step"],2), {k:round(v,2) for k,v in r["stage_ms"].items()}, r["diagnostics"], "oracle", p["oracle_sample_searches"], "idx_err", p.get("max_idx_err_vs_oracle_sample"), "score", p.get("max_score_err_over_tolerance_vs_oracle_sample"), "planted", p["max_shift_err_samples_vs_planted"], "ws", r["workspace_bytes"])
    except Exception as e: print(f, "ERR", e)
PY
