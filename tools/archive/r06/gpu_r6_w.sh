# plan made once (kept from sushi_hip_batch_bytes for sushi_hip_batch_create), its schedule by a counting sort: the set-up and the per-call costs
O=gpurun_out/r06w
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export SUSHI_BENCH_CACHE=/tmp/sushi_bench_cache
timeout 900 python -m pytest tests/test_native_abi.py tests/test_gpu_parity.py tests/test_pair_exclusion.py -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
for rep in 1 2; do
timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --emulate-shards 0 > $O/b.json 2> $O/b.err
python -c "
import json; d=json.load(open('$O/b.json')); print('cfg2', round(d['ms_per_step'],3), d['setup_ms'], round(d.get('one_shot_events_per_s')), round(d.get('one_shot_incl_process_start_events_per_s')))"
done
timeout 400 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --emulate-shards 0 --config 4 > $O/b.json 2> $O/b.err
python -c "
import json; d=json.load(open('$O/b.json')); print('cfg4', round(d['ms_per_step'],3), d['setup_ms'], round(d.get('one_shot_events_per_s')))"
timeout 120 python tools/call_breakdown.py | tee $O/call_breakdown.json
timeout 120 python tools/latency.py > $O/latency.json 2> $O/latency.err
python -c "
import json; d=json.load(open('$O/latency.json'))
for k,v in d.items(): print(k, {a: round(b,4) for a,b in v.items() if not isinstance(b, dict)})"
