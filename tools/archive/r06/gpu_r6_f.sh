# tspec ablations (garbage results: stage times only)
O=gpurun_out/r06f
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export SUSHI_BENCH_CACHE=/tmp/sushi_bench_cache
timeout 400 python bench.py --steps 3 --warmup 1 --profile-only --emulate-shards 0 > $O/warm.json 2> $O/warm.err
timeout 200 python tools/stage_times.py --steps 5 --tag product | tee -a $O/tspec_ablations.jsonl
for v in nofft nohandover nostore nolow; do
  SUSHI_HIP_LIB=$PWD/sushi_amd/lib/libsushi_hip_ta_$v.so timeout 300 python tools/stage_times.py --steps 1 --tag $v | tee -a $O/tspec_ablations.jsonl
done
for wl in "dub_cc:--source dub --method ccoeff_normed" "dub_whole:--source dub --exclusion whole" "dub_band:--source dub --exclusion band"; do
  name=${wl%%:*}; args=${wl#*:}
  timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --emulate-shards 0 $args > $O/bench_$name.json 2> $O/bench_$name.err || tail -3 $O/bench_$name.err
done
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/bench_*.json")):
    try:
        d=json.load(open(f)); r=d["roofline"]; g=r.get("diagnostics") or {}; p=d["parity"]
        print(f.split("/")[-1], round(d["value"]), round(d["ms_per_step"],3), {k: round(v,2) for k,v in (r.get("stage_ms") or {}).items()}, "pairs", g.get("pairs_transformed"), "band", g.get("band"), "idx_err", p.get("max_idx_err_vs_oracle_sample"))
    except Exception as e: print(f, "ERR", e)
PY
