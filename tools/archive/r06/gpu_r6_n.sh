# lanes: whole-row forms on one stream; the shard emulation under lanes
O=gpurun_out/r06n
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export SUSHI_BENCH_CACHE=/tmp/sushi_bench_cache
line() {   # name lanes args...
  name=$1; L=$2; shift 2
  if [ $L = auto ]; then unset SUSHI_HIP_LANES; else export SUSHI_HIP_LANES=$L; fi
  timeout 600 python bench.py --steps 10 --warmup 3 --profile-only --emulate-shards 0 "$@" > $O/b.json 2> $O/b.err
  python - <<PY | tee -a $O/sweep.txt
import json
try:
    d=json.load(open("$O/b.json")); r=d["roofline"]; g=r.get("diagnostics") or {}
    print("$name", "$L", round(d["ms_per_step"],3), {k: round(v,2) for k,v in (r.get("stage_ms") or {}).items()}, g.get("pairs_transformed"), d["parity"].get("max_shift_err_samples_vs_planted"), d["parity"].get("events_beyond_one_sample_of_planted"))
except Exception as e:
    print("$name", "$L", "FAILED", e, open("$O/b.err").read()[-600:])
PY
  unset SUSHI_HIP_LANES
}
line cfg2 auto
for L in 1:1 2:2 4:2; do line ev400 $L --events 400; line ev800 $L --events 800; done
for L in 1:1 auto 3:1; do line unrelated $L --unrelated; done
for L in 1:1 auto 3:1; do line snr0 $L --snr 0; done
for L in 1:1 auto; do line whole $L --exclusion whole; done
for L in 1:1 2:2 4:2 6:3 9:3; do
  export SUSHI_HIP_LANES=$L
  timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --emulate-shards 8 > $O/b.json 2> $O/b.err
  python -c "
import json
d=json.load(open('$O/b.json')); se=d.get('shard_emulation') or {}
print('shards', '$L', {g: (v.get('slowest_shard_ms'), v.get('mean_shard_ms')) for g, v in (se.get('by_world_size') or {}).items()})" | tee -a $O/sweep.txt
done
