set -x
O=gpurun_out/r03i
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export SUSHI_BENCH_CACHE=/tmp/sushi_bench_cache
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --cpu-sample 512 > $O/bench_cfg2_n1.json 2> $O/b.err; tail -2 $O/b.err
for v in prev yhalf product prev yhalf product; do
  if [ $v = product ]; then unset SUSHI_HIP_LIB; else export SUSHI_HIP_LIB=$PWD/sushi_amd/lib/libsushi_hip_$v.so; fi
  timeout 120 python tools/stage_times.py --steps 10 --tag $v 2>/dev/null | tail -1 | tee -a $O/ab.log
done
unset SUSHI_HIP_LIB
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_ccoeff.py -m gpu -q > $O/pytest_gpu.log 2>&1; tail -12 $O/pytest_gpu.log
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r03i/*.json")):
    try:
        d=json.load(open(f)); r=d["roofline"]; p=d["parity"]
        print(f.split("/")[-1], round(d["value"]), round(d["ms_per_step"],2), {k:round(v,2) for k,v in r["stage_ms"].items()}, r["diagnostics"], "oracle", p["oracle_sample_searches"], "idx_err", p.get("max_idx_err_vs_oracle_sample"), "score", p.get("max_score_err_over_tolerance_vs_oracle_sample"), "planted", p["max_shift_err_samples_vs_planted"])
    except Exception as e: print(f, "ERR", e)
PY
