# Round-6 evidence (profiles/r06/final/): per workload -- kernel trace, FETCH_SIZE pass, WRITE_SIZE pass, profiles/pmc_traffic.json entry
# (bytes PER STEP: PMC_RUNS), then the bench line (which then carries roofline.traffic).
# WL="name:key-suffix:bench args|..." ; WITH_TESTS=1 adds the whole GPU suite.
set -x
O=gpurun_out/${OUT:-r06final}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export SUSHI_BENCH_CACHE=/tmp/sushi_bench_cache
run_wl() {
  name=$1; key=$2; args=$3; steps=${4:-5}
  B="python bench.py --steps $steps --warmup 1 --profile-only --emulate-shards 0 $args"            # (no forked oracle workers under the profiler)
  timeout 400 $B > $O/warm_$name.json 2> $O/warm_$name.err            # builds the stream cache
  rm -rf gpurun_out/prof_*
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_kt -o kt -- $B > $O/kt_$name.log 2>&1
  cp gpurun_out/prof_kt/kt_kernel_stats.csv $O/${name}_kernel_stats.csv
  if [ $name = cfg2 ]; then
    # the same workload with its sub-batches one after the other on one stream: per-kernel times that add up to the step
    rm -rf gpurun_out/prof_kt1
    SUSHI_HIP_LANES=1:1 timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_kt1 -o kt -- $B > $O/kt_${name}_lanes1.log 2>&1
    cp gpurun_out/prof_kt1/kt_kernel_stats.csv $O/${name}_lanes1_kernel_stats.csv
  fi
  timeout 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof_fetch -o fetch -- $B > $O/fetch_$name.log 2>&1
  timeout 150 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/prof_write -o write -- $B > $O/write_$name.log 2>&1 || echo "WRITE_SIZE pass of $name cut by its timeout" | tee -a $O/notes.txt
  python tools/summarize_pmc.py $O/${name}_pmc_summary.csv $(find gpurun_out/prof_fetch gpurun_out/prof_write -name '*counter_collection.csv' 2>/dev/null)
  PMC_RUNS=$((steps + 2)) python tools/make_pmc_traffic.py $O/${name}_pmc_summary.csv profiles/pmc_traffic.json "$key" "$COMMIT" > /dev/null || echo "no pmc_traffic entry for $name" | tee -a $O/notes.txt
  head -10 $O/${name}_kernel_stats.csv | cut -c1-150
}
IFS='|' read -ra W <<< "$WL"
for w in "${W[@]}"; do
  name=$(echo "$w" | cut -d: -f1); key=$(echo "$w" | cut -d: -f2); args=$(echo "$w" | cut -d: -f3-)
  run_wl "$name" "$key" "$args"
  case "$name" in
    cfg2) timeout 400 python bench.py --steps 20 --warmup 5 > $O/bench_${name}_n1.json 2> $O/bench_${name}_n1.err ;;
    cfg4) timeout 900 python bench.py --steps 5 --warmup 2 $args --cpu-sample 256 --emulate-shards 0 > $O/bench_${name}_n1.json 2> $O/bench_${name}_n1.err ;;
    *) timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --emulate-shards 0 $args > $O/bench_${name}_n1.json 2> $O/bench_${name}_n1.err ;;
  esac
done
# bench lines without counter passes: the source's noise level, unrelated audio, the forms of the exclusion side by side
IFS='|' read -ra X <<< "$EXTRA"
for w in "${X[@]}"; do
  name=$(echo "$w" | cut -d: -f1); args=$(echo "$w" | cut -d: -f2-)
  case "$name" in
    lanes1*) SUSHI_HIP_LANES=1:1 timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --emulate-shards 0 --cpu-sample 128 $args > $O/bench_${name}_n1.json 2> $O/bench_${name}_n1.err ;;
    stat*) SUSHI_HIP_BOUND_MODEL=statistical timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --emulate-shards 0 --cpu-sample 128 $args > $O/bench_${name}_n1.json 2> $O/bench_${name}_n1.err ;;
    *) timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --emulate-shards 0 --cpu-sample 128 $args > $O/bench_${name}_n1.json 2> $O/bench_${name}_n1.err ;;
  esac
done
cp profiles/pmc_traffic.json $O/pmc_traffic.json
if [ "$WITH_TESTS" = 1 ]; then
timeout 1200 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
fi
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $O/smoke.log
timeout 60 python tools/kernel_resources.py sushi_fft > $O/kernel_resources_fft.txt 2>&1
timeout 400 python tools/kernel_resources.py sushi_hip > $O/kernel_resources_hip.txt 2>&1
timeout 120 python tools/latency.py > $O/latency.json 2> $O/latency.err
timeout 120 python tools/call_breakdown.py > $O/call_breakdown.json 2> $O/call_breakdown.err
if [ "$WITH_HUNTS" = 1 ]; then
timeout 900 python tools/excluded_audit_hunt.py > $O/excluded_audit_hunt.jsonl 2> $O/excluded_audit_hunt.err; tail -1 $O/excluded_audit_hunt.jsonl
timeout 600 python tools/bound_hunt.py 126 > $O/bound_hunt.txt 2>&1; tail -3 $O/bound_hunt.txt
fi
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/bench_*.json")):
    try:
        d=json.load(open(f)); r=d["roofline"]; p=d["parity"]; g=r.get("diagnostics") or {}
        print(f.split("/")[-1], round(d["value"]), round(d["ms_per_step"],2), {k: round(v,2) for k,v in (r.get("stage_ms") or {}).items()}, "frac", round(r["frac"],3), "traffic", r.get("traffic") and round(r["traffic"]), "x_alg", r.get("step_traffic_over_algorithmic") and round(r["step_traffic_over_algorithmic"],2), "pairs", g.get("pairs_transformed"), "band", g.get("band"), "susp", g.get("suspended"), "oracle", p["oracle_sample_searches"], "idx_err", p.get("max_idx_err_vs_oracle_sample"), "cpu", d["cpu_baseline"] and round(d["cpu_baseline"]["value"],1), "one_shot", d.get("one_shot_events_per_s") and round(d["one_shot_events_per_s"]), "aud", g.get("excluded_audited"), "slb", g.get("max_slb_ratio_excluded") and round(g.get("max_slb_ratio_excluded"),3), "ratios", g.get("max_bound_ratio") and round(g.get("max_bound_ratio"),3), g.get("max_bound_ratio_noncandidate") and round(g.get("max_bound_ratio_noncandidate"),3))
        se=d.get("shard_emulation")
        if se:
            for k,v in se["by_world_size"].items(): print("   shards G=%s max %.3f ms mean %.3f speedup %.2f" % (k, v["max_shard_ms"], v["mean_shard_ms"], v["implied_speedup_over_one_gpu"]))
    except Exception as e: print(f, "ERR", e)
PY
