# Round 6, first call: cv2 probe, the new tests, worst-case vs statistical bound A/B at 20 / 12 / 6 dB, the default bench line.
set -x
O=gpurun_out/r06a
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export SUSHI_BENCH_CACHE=/tmp/sushi_bench_cache
bash tools/cv2_probe.sh > $O/cv2_probe_tail.txt 2>&1; cp gpurun_out/cv2_probe.txt $O/ 2>/dev/null
timeout 900 python -m pytest tests/test_pair_exclusion.py tests/test_bound_stress.py tests/test_distributed_gpu.py tests/test_native_abi.py -m gpu -x -q > $O/pytest_new.log 2>&1; tail -5 $O/pytest_new.log
for snr in 20 12 6; do
  for model in worst_case statistical; do
    SUSHI_HIP_BOUND_MODEL=$model timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --emulate-shards 0 --snr $snr > $O/bench_snr${snr}_${model}.json 2> $O/bench_snr${snr}_${model}.err
  done
done
timeout 600 python bench.py > $O/bench_cfg2_default.json 2> $O/bench_cfg2_default.err
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/bench_*.json")):
    try:
        d=json.load(open(f)); r=d["roofline"]; g=r.get("diagnostics") or {}
        print(f.split("/")[-1], round(d["value"]), round(d["ms_per_step"],3), {k: round(v,2) for k,v in (r.get("stage_ms") or {}).items()}, "pairs", g.get("pairs_transformed"), "aud", g.get("excluded_audited"), "ratio", g.get("max_slb_ratio_excluded"), "viol", g.get("slb_violations"), "idx_err", d["parity"].get("max_idx_err_vs_oracle_sample"))
        se=d.get("shard_emulation")
        if se:
            for k,v in se["by_world_size"].items(): print("  G=%s max %.3f mean %.3f speedup %.2f" % (k, v["max_shard_ms"], v["mean_shard_ms"], v["implied_speedup_over_one_gpu"]), [x["host_launch_ms"] for x in v["shards"]][:2])
    except Exception as e: print(f, "ERR", e)
PY
