# Round 6: after the row-energy fix -- the tone-burst probes, the whole GPU suite, the bench A/B of the bound models
set -x
O=gpurun_out/r06c
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export SUSHI_BENCH_CACHE=/tmp/sushi_bench_cache
python tools/fs8_probe2.py 2>&1 | grep -v amdgpu.ids > $O/fs8_probe2.txt; cat $O/fs8_probe2.txt
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; tail -5 $O/pytest_gpu.log
for snr in 20 6; do
  for model in worst_case statistical; do
    SUSHI_HIP_BOUND_MODEL=$model timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --emulate-shards 0 --snr $snr > $O/bench_snr${snr}_${model}.json 2> $O/bench_snr${snr}_${model}.err
  done
done
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/bench_*.json")):
    try:
        d=json.load(open(f)); r=d["roofline"]; g=r.get("diagnostics") or {}
        print(f.split("/")[-1], round(d["value"]), round(d["ms_per_step"],3), {k: round(v,2) for k,v in (r.get("stage_ms") or {}).items()}, "pairs", g.get("pairs_transformed"), "cand", g.get("candidates"), "ratios", round(g.get("max_bound_ratio"),3), round(g.get("max_bound_ratio_noncandidate"),3), "flagged", g.get("flagged"), "idx_err", d["parity"].get("max_idx_err_vs_oracle_sample"))
    except Exception as e: print(f, "ERR", e)
PY
