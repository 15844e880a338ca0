# experiment: stream priorities of the lanes, an occupancy cap on mac_kernel<1024>
O=gpurun_out/r06p
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export SUSHI_BENCH_CACHE=/tmp/sushi_bench_cache
timeout 400 python bench.py --steps 3 --warmup 1 --profile-only --emulate-shards 0 > $O/warm.json 2> $O/warm.err
for rep in 1 2; do
for V in "0 0" "0 1" "0 2" "0 3" "20000 0" "20000 1" "45000 0"; do
  set -- $V
  export SUSHI_HIP_EXP_MAC_LDS=$1 SUSHI_HIP_EXP_PRIO=$2
  for L in auto 2:2; do
  if [ $L = auto ]; then unset SUSHI_HIP_LANES; else export SUSHI_HIP_LANES=$L; fi
  timeout 200 python bench.py --steps 20 --warmup 3 --profile-only --emulate-shards 0 > $O/b.json 2> $O/b.err
  python - <<PY | tee -a $O/sweep.txt
import json
d=json.load(open("$O/b.json")); r=d["roofline"]
print("lds=$1 prio=$2 lanes=$L", round(d["ms_per_step"],3), {k: round(v,2) for k,v in (r.get("stage_ms") or {}).items()})
PY
  done
done
done
