# Round 3: the wave-plan inverse transform (fft_core.hpp) against the previous commit's library on one box, then parity.
set -x
O=gpurun_out/r03e
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export SUSHI_BENCH_CACHE=/tmp/sushi_bench_cache
timeout 300 python bench.py --steps 20 --warmup 5 > $O/bench_cfg2_n1.json 2> $O/bench_cfg2_n1.err; head -c 300 $O/bench_cfg2_n1.json; echo; tail -3 $O/bench_cfg2_n1.err
for v in prev product prev product; do
  if [ $v = prev ]; then export SUSHI_HIP_LIB=$PWD/sushi_amd/lib/libsushi_hip_prev.so; else unset SUSHI_HIP_LIB; fi
  timeout 120 python tools/stage_times.py --steps 10 --tag $v 2>/dev/null | tail -1 | tee -a $O/ab.log
done
unset SUSHI_HIP_LIB
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_ccoeff.py tests/test_shifts.py -m gpu -q -x > $O/pytest_gpu.log 2>&1; tail -5 $O/pytest_gpu.log
python - <<'PY'
import json
d=json.load(open("gpurun_out/r03e/bench_cfg2_n1.json"))
print(d["value"], d["roofline"]["stage_ms"], d["roofline"]["diagnostics"], d["parity"], d["cpu_baseline"], d["setup_ms"])
PY
