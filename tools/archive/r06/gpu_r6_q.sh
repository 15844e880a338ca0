# experiment: bound_low_kernel's grid as a multiple of what is resident (persistent waves against the hardware's own balancing), with and without lanes
O=gpurun_out/r06q
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export SUSHI_BENCH_CACHE=/tmp/sushi_bench_cache
timeout 400 python bench.py --steps 3 --warmup 1 --profile-only --emulate-shards 0 > $O/warm.json 2> $O/warm.err
for rep in 1 2; do
for G in 1 2 4 8 16; do
  export SUSHI_HIP_EXP_GRID=$G
  for L in auto 1:1; do
  if [ $L = auto ]; then unset SUSHI_HIP_LANES; else export SUSHI_HIP_LANES=$L; fi
  timeout 200 python bench.py --steps 20 --warmup 3 --profile-only --emulate-shards 0 > $O/b.json 2> $O/b.err
  python - <<PY | tee -a $O/sweep.txt
import json
d=json.load(open("$O/b.json")); r=d["roofline"]
print("grid=$G lanes=$L", round(d["ms_per_step"],3), {k: round(v,2) for k,v in (r.get("stage_ms") or {}).items()})
PY
  done
done
done
