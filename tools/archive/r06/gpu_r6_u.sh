# experiment: how many listed pairs the would-be dense searches must hold before the dense form is taken
O=gpurun_out/r06u
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export SUSHI_BENCH_CACHE=/tmp/sushi_bench_cache
line() {   # name args...
  name=$1; shift 1
  timeout 600 python bench.py --steps 10 --warmup 3 --profile-only --emulate-shards 0 "$@" > $O/b.json 2> $O/b.err
  python - <<PY | tee -a $O/sweep.txt
import json
try:
    d=json.load(open("$O/b.json")); r=d["roofline"]; g=r.get("diagnostics") or {}
    print("min=$SUSHI_HIP_EXP_DENSE_MIN lanes=$SUSHI_HIP_LANES", "$name", round(d["ms_per_step"],3), {k: round(v,2) for k,v in (r.get("stage_ms") or {}).items()}, g.get("pairs_transformed"))
except Exception as e:
    print("$name", "FAILED", e, open("$O/b.err").read()[-600:])
PY
}
for M in 0 2048 8192 32768 1000000000; do
export SUSHI_HIP_EXP_DENSE_MIN=$M
for L in 1:1 auto; do
if [ $L = auto ]; then unset SUSHI_HIP_LANES; else export SUSHI_HIP_LANES=$L; fi
line cfg2
line snr6 --snr 6
line dub --source dub
line partial --source partial
done
done
