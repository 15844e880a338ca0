# Round 3, second GPU call: CCOEFF on the FFT path, two-sided bound check, reworked oracle leg on the 256-core box.
set -x
O=gpurun_out/r03b
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export SUSHI_BENCH_CACHE=/tmp/sushi_bench_cache
timeout 900 python -m pytest tests/test_ccoeff.py tests/test_gpu_parity.py tests/test_cv2_crosscheck.py tests/test_bench_contract.py -m gpu -q > $O/pytest_gpu.log 2>&1; tail -15 $O/pytest_gpu.log
timeout 300 python bench.py --steps 20 --warmup 5 > $O/bench_cfg2_n1.json 2> $O/bench_cfg2_n1.err; head -c 300 $O/bench_cfg2_n1.json; echo; tail -3 $O/bench_cfg2_n1.err
timeout 300 python bench.py --steps 10 --warmup 3 --method ccoeff_normed --no-cpu-baseline > $O/bench_cfg2_ccoeff_n1.json 2> $O/bench_cfg2_ccoeff_n1.err; head -c 300 $O/bench_cfg2_ccoeff_n1.json; echo; tail -3 $O/bench_cfg2_ccoeff_n1.err
python - <<'PY'
import json
for f in ("bench_cfg2_n1","bench_cfg2_ccoeff_n1"):
    try:
        d=json.load(open("gpurun_out/r03b/%s.json"%f))
        print(f, d["value"], d["roofline"]["stage_ms"], d["roofline"]["diagnostics"], d["parity"], d["cpu_baseline"], d["setup_ms"])
    except Exception as e: print(f, "ERR", e)
PY
