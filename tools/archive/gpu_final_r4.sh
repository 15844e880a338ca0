# Round-4 evidence (profiles/r04/final/): per workload -- kernel trace, FETCH_SIZE pass, WRITE_SIZE pass (alone, short timeout:
# it hung in round 3), profiles/pmc_traffic.json entry, then the bench line (which then carries roofline.traffic).
# WL="name:key-suffix:bench args|..." ; WITH_TESTS=1 adds the whole GPU suite; WITH_TCP=1 the L1 request counters of the default workload.
set -x
O=gpurun_out/${OUT:-r04final}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export SUSHI_BENCH_CACHE=/tmp/sushi_bench_cache
run_wl() {
  name=$1; key=$2; args=$3; steps=${4:-5}
  B="python bench.py --steps $steps --warmup 1 --profile-only $args"            # (no forked oracle workers under the profiler)
  timeout 400 $B > $O/warm_$name.json 2> $O/warm_$name.err            # builds the stream cache
  rm -rf gpurun_out/prof_*
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_kt -o kt -- $B > $O/kt_$name.log 2>&1
  cp gpurun_out/prof_kt/kt_kernel_stats.csv $O/${name}_kernel_stats.csv
  timeout 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof_fetch -o fetch -- $B > $O/fetch_$name.log 2>&1
  timeout 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/prof_write -o write -- $B > $O/write_$name.log 2>&1 || echo "WRITE_SIZE pass of $name cut by its timeout" | tee -a $O/notes.txt
  python tools/summarize_pmc.py $O/${name}_pmc_summary.csv $(find gpurun_out/prof_fetch gpurun_out/prof_write -name '*counter_collection.csv' 2>/dev/null)
  python tools/make_pmc_traffic.py $O/${name}_pmc_summary.csv profiles/pmc_traffic.json "$key" "$COMMIT" > /dev/null || echo "no pmc_traffic entry for $name" | tee -a $O/notes.txt
  head -8 $O/${name}_kernel_stats.csv | cut -c1-150
}
IFS='|' read -ra W <<< "$WL"
for w in "${W[@]}"; do
  name=$(echo "$w" | cut -d: -f1); key=$(echo "$w" | cut -d: -f2); args=$(echo "$w" | cut -d: -f3-)
  run_wl "$name" "$key" "$args"
  case "$name" in
    cfg2) timeout 400 python bench.py --steps 20 --warmup 5 > $O/bench_${name}_n1.json 2> $O/bench_${name}_n1.err ;;
    cfg4) timeout 900 python bench.py --steps 5 --warmup 2 $args --cpu-sample 256 > $O/bench_${name}_n1.json 2> $O/bench_${name}_n1.err ;;
    *) timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline $args > $O/bench_${name}_n1.json 2> $O/bench_${name}_n1.err ;;
  esac
done
if [ "$WITH_TCP" = 1 ]; then
timeout 200 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum --kernel-include-regex "ifft_kernel|mac_kernel|bound_kernel" --output-format csv -d gpurun_out/prof_tcp -o tcp -- python tools/stage_times.py --steps 2 --tag tcp > $O/tcp.log 2>&1
python tools/summarize_pmc.py $O/cfg2_tcp_summary.csv $(find gpurun_out/prof_tcp -name '*counter_collection.csv'); cat $O/cfg2_tcp_summary.csv
fi
cp profiles/pmc_traffic.json $O/pmc_traffic.json
if [ "$WITH_TESTS" = 1 ]; then
timeout 1200 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
fi
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $O/smoke.log
timeout 60 python tools/kernel_resources.py sushi_fft > $O/kernel_resources_fft.txt 2>&1
timeout 60 python tools/kernel_resources.py sushi_hip > $O/kernel_resources_hip.txt 2>&1
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/bench_*.json")):
    try:
        d=json.load(open(f)); r=d["roofline"]; p=d["parity"]
        print(f.split("/")[-1], round(d["value"]), round(d["ms_per_step"],2), {k: round(v,2) for k,v in (r.get("stage_ms") or {}).items()}, "kernel", r.get("kernel"), "frac", round(r["frac"],3), "traffic", r.get("traffic"), "oracle", p["oracle_sample_searches"], "idx_err", p.get("max_idx_err_vs_oracle_sample"), "cpu", d["cpu_baseline"] and round(d["cpu_baseline"]["value"],1), "one_shot", d.get("one_shot_events_per_s") and round(d["one_shot_events_per_s"]))
    except Exception as e: print(f, "ERR", e)
PY
