# lanes on the other workloads: SUSHI_HIP_LANES 1:1 against the default (and a few more), one line per run
O=gpurun_out/r06m
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
for L in 1:1 auto; do line cfg2 $L; done
for L in 1:1 2:2 3:3 4:2 6:3; do line cfg1 $L --config 1; done
for L in 1:1 auto 12:4; do line cfg4 $L --config 4; done
for L in 1:1 auto; do line dub $L --source dub; done
for L in 1:1 auto; do line partial $L --source partial; done
for L in 1:1 auto; do line unrelated $L --unrelated; done
for L in 1:1 auto; do line hard $L --hard-frac 0.05; done
for L in 1:1 auto; do line ccoeff $L --method ccoeff_normed; done
for L in 1:1 auto; do line u8 $L --sample-type uint8; done
for L in 1:1 auto; do line snr6 $L --snr 6; done
for L in 1:1 auto; do line snr0 $L --snr 0; done
# a shard of the 8-rank run (375 events): the emulation under three settings
for L in 1:1 2:2 3:3; do
  export SUSHI_HIP_LANES=$L
  timeout 300 python bench.py --steps 5 --warmup 2 --profile-only --emulate-shards 8 > $O/b.json 2> $O/b.err
  python -c "
import json
d=json.load(open('$O/b.json')); print('shards8', '$L', json.dumps(d.get('shard_emulation'))[:600])" | tee -a $O/sweep.txt
done
