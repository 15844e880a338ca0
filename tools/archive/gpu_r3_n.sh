# the last GPU-minute of the round: operand maps of v_mfma_f32_16x16x32_f16, then libsushi_hip_mfma.so (first pass of the inverse
# transforms on the matrix pipe; tools/experiments/r04_ifft_mfma_first_pass.patch) against the product: one bench line with its
# planted-offset check and a 64-search oracle sample, and stage times
O=gpurun_out/r3n
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export SUSHI_BENCH_CACHE=/tmp/sushi_bench_cache
timeout 10 tools/ubench/mfma_layout | tee $O/mfma_layout.log
SUSHI_HIP_LIB=$PWD/sushi_amd/lib/libsushi_hip_mfma.so timeout 40 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --cpu-sample 64 > $O/bench_mfma.json 2> $O/b.err; tail -2 $O/b.err
for v in product mfma; do
  if [ $v = product ]; then unset SUSHI_HIP_LIB; else export SUSHI_HIP_LIB=$PWD/sushi_amd/lib/libsushi_hip_$v.so; fi
  timeout 15 python tools/stage_times.py --steps 10 --tag $v 2>/dev/null | tail -1 | tee -a $O/ab.log
done
python - <<'PY'
import json
d=json.load(open("gpurun_out/r3n/bench_mfma.json")); r=d["roofline"]; p=d["parity"]
print(round(d["value"]), round(d["ms_per_step"],2), r["stage_ms"], r["diagnostics"], p["oracle_sample_searches"], p.get("max_idx_err_vs_oracle_sample"), p.get("max_abs_score_err_vs_oracle_sample"), p["max_shift_err_samples_vs_planted"])
PY
