# the first run's transform launch sized for "nothing known yet": first_step_ms by workload
O=gpurun_out/r06y
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export SUSHI_BENCH_CACHE=/tmp/sushi_bench_cache
for rep in 1 2; do
for W in "cfg2:" "dub:--source dub" "partial:--source partial" "snr6:--snr 6" "cc:--method ccoeff_normed" "unrelated:--unrelated"; do
  name=${W%%:*}; args=${W#*:}
  timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --cpu-sample 64 --emulate-shards 0 $args > $O/b.json 2> $O/b.err
  python -c "
import json; d=json.load(open('$O/b.json')); print('$name', 'first', round(d['first_step_ms'],2), 'steady', round(d['ms_per_step'],2), 'one_shot', round(d['one_shot_events_per_s']), d['parity'].get('max_idx_err_vs_oracle_sample'))" | tee -a $O/sweep.txt
done
done
