# experiment: sub-batches of a lane that shrink from round to round (a shorter ragged end), lanes that start out of step
O=gpurun_out/r06r
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export SUSHI_BENCH_CACHE=/tmp/sushi_bench_cache
timeout 400 python bench.py --steps 3 --warmup 1 --profile-only --emulate-shards 0 > $O/warm.json 2> $O/warm.err
for rep in 1 2; do
for T in 1.0:0.0 0.7:0.0 0.5:0.0 1.0:0.2 0.7:0.2 0.5:0.3 0.35:0.0; do
  export SUSHI_HIP_EXP_TAPER=$T
  for L in 9:3 12:3 6:3 6:2; do
  export SUSHI_HIP_LANES=$L
  timeout 200 python bench.py --steps 20 --warmup 3 --profile-only --emulate-shards 0 > $O/b.json 2> $O/b.err
  python - <<PY | tee -a $O/sweep.txt
import json
d=json.load(open("$O/b.json")); r=d["roofline"]
print("taper=$T lanes=$L", round(d["ms_per_step"],3))
PY
  done
done
done
