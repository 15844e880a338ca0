# A/B of sushi_amd/lib/libsushi_hip_prev.so (a build of an earlier commit) against the product library on one box, then parity
set -x
O=gpurun_out/ab2
mkdir -p $O; rm -f $O/ab.log
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export SUSHI_BENCH_CACHE=/tmp/sushi_bench_cache
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --cpu-sample 256 > $O/bench_cfg2_n1.json 2> $O/b.err; tail -2 $O/b.err
for v in prev product prev product; do
  if [ $v = product ]; then unset SUSHI_HIP_LIB; else export SUSHI_HIP_LIB=$PWD/sushi_amd/lib/libsushi_hip_$v.so; fi
  timeout 120 python tools/stage_times.py --steps 10 --tag $v 2>/dev/null | tail -1 | tee -a $O/ab.log
done
unset SUSHI_HIP_LIB
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_ccoeff.py -m gpu -q > $O/pytest_gpu.log 2>&1; tail -4 $O/pytest_gpu.log
python - <<'PY'
import json
d=json.load(open("gpurun_out/ab2/bench_cfg2_n1.json")); r=d["roofline"]; p=d["parity"]
print(round(d["value"]), round(d["ms_per_step"],2), r["stage_ms"], r["diagnostics"], p["oracle_sample_searches"], p.get("max_idx_err_vs_oracle_sample"))
PY
