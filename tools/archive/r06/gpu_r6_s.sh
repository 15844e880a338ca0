# the early records of a drop-in call; the one-sub-batch cut made by the first run that wants it
O=gpurun_out/r06s
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export SUSHI_BENCH_CACHE=/tmp/sushi_bench_cache
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_pair_exclusion.py tests/test_native_abi.py tests/test_host_wav.py tests/test_shifts.py tests/test_shifts_golden.py -m gpu -x -q > $O/pytest.log 2>&1; tail -5 $O/pytest.log
for rep in 1 2; do
timeout 120 python tools/latency.py > $O/latency$rep.json 2> $O/latency.err
timeout 120 python tools/call_breakdown.py | tee $O/call_breakdown$rep.json
python -c "
import json; d=json.load(open('$O/latency$rep.json'))
for k,v in d.items(): print(k, {a: round(b,4) for a,b in v.items() if not isinstance(b, dict)})"
done
timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --emulate-shards 0 > $O/b.json 2> $O/b.err
python -c "
import json; d=json.load(open('$O/b.json')); print('cfg2', d['ms_per_step'], d['setup_ms'], d.get('one_shot_events_per_s'))"
timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --emulate-shards 0 --unrelated > $O/b.json 2> $O/b.err
python -c "
import json; d=json.load(open('$O/b.json')); print('unrelated', d['ms_per_step'], d['setup_ms'], d['parity'].get('max_idx_err_vs_oracle_sample'))"
timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --emulate-shards 0 --exclusion whole > $O/b.json 2> $O/b.err
python -c "
import json; d=json.load(open('$O/b.json')); print('whole', d['ms_per_step'], d['setup_ms'], d['parity'].get('max_idx_err_vs_oracle_sample'))"
