# Packed-half block / pattern spectra + v_dot2_f32_f16 multiply-accumulate (4 bins per lane): product against
# libsushi_hip_prev.so (the float32-operand kernels of commit 524a7f2) and a 16-searches-per-wave build, then the GPU suite
set -x
O=gpurun_out/r3j
mkdir -p $O; rm -f $O/ab.log
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export SUSHI_BENCH_CACHE=/tmp/sushi_bench_cache
timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --cpu-sample 256 > $O/bench_cfg2_n1.json 2> $O/b.err; tail -3 $O/b.err
for v in prev product s16 prev product s16; do
  if [ $v = product ]; then unset SUSHI_HIP_LIB; else export SUSHI_HIP_LIB=$PWD/sushi_amd/lib/libsushi_hip_$v.so; fi
  timeout 120 python tools/stage_times.py --steps 10 --tag $v 2>/dev/null | tail -1 | tee -a $O/ab.log
done
unset SUSHI_HIP_LIB
timeout 900 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; tail -40 $O/pytest_gpu.log
python - <<'PY'
import json
d=json.load(open("gpurun_out/r3j/bench_cfg2_n1.json")); r=d["roofline"]; p=d["parity"]
print(round(d["value"]), round(d["ms_per_step"],2), r["stage_ms"], r["diagnostics"], p["oracle_sample_searches"], p.get("max_idx_err_vs_oracle_sample"), p.get("max_abs_score_err_vs_oracle_sample"))
PY
