# Round 6: per-search dense selection -- the dub / low-SNR / unrelated lines, the default, the exclusion tests
set -x
O=gpurun_out/r06h
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export SUSHI_BENCH_CACHE=/tmp/sushi_bench_cache
timeout 900 python -m pytest tests/test_pair_exclusion.py tests/test_bound_stress.py tests/test_gpu_parity.py tests/test_ccoeff.py -m gpu -q -x > $O/pytest_sel.log 2>&1; tail -3 $O/pytest_sel.log
for wl in "cfg2:" "dub:--source dub" "dub_cc:--source dub --method ccoeff_normed" "dub_band:--source dub --exclusion band" "snr0:--snr 0" "snr0_band:--snr 0 --exclusion band" "snr6:--snr 6" "unrelated:--unrelated"; do
  name=${wl%%:*}; args=${wl#*:}
  timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --emulate-shards 0 $args > $O/bench_$name.json 2> $O/bench_$name.err || tail -3 $O/bench_$name.err
done
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/bench_*.json")):
    try:
        d=json.load(open(f)); r=d["roofline"]; g=r.get("diagnostics") or {}; p=d["parity"]
        print(f.split("/")[-1], round(d["value"]), round(d["ms_per_step"],3), {k: round(v,2) for k,v in (r.get("stage_ms") or {}).items()}, "pairs", g.get("pairs_transformed"), "band", g.get("band"), "susp", g.get("suspended"), "flagged", g.get("flagged"), "oracle", p["oracle_sample_searches"], "idx_err", p.get("max_idx_err_vs_oracle_sample"))
    except Exception as e: print(f, "ERR", e)
PY
