# experiment: a two-stage pipeline (bound stages on one stream, listed pairs' stages on a second) against three identical lanes
O=gpurun_out/r06pipe
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export SUSHI_BENCH_CACHE=/tmp/sushi_bench_cache
SUSHI_HIP_PIPE=1 timeout 600 python -m pytest tests/test_pair_exclusion.py -m gpu -x -q -k "lanes or alternate or by_itself" > $O/pytest.log 2>&1; tail -2 $O/pytest.log
timeout 400 python bench.py --steps 3 --warmup 1 --profile-only --emulate-shards 0 > /dev/null 2>&1
for rep in 1 2; do
for P in 0 1; do
for L in 9:3 12:3 18:3 6:2 12:2 12:4 24:4; do
  export SUSHI_HIP_PIPE=$P SUSHI_HIP_LANES=$L
  timeout 200 python bench.py --steps 20 --warmup 3 --profile-only --emulate-shards 0 > $O/b.json 2> $O/b.err
  python -c "
import json; d=json.load(open('$O/b.json')); print('pipe=$P lanes=$L', round(d['ms_per_step'],3), d['parity'].get('events_beyond_one_sample_of_planted'))" | tee -a $O/sweep.txt
done
done
done
