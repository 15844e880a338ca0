# Round 6: what Sushi's inputs look like -- another encode, a dub -- and the low-SNR / unrelated lines again
set -x
O=gpurun_out/r06e
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export SUSHI_BENCH_CACHE=/tmp/sushi_bench_cache
for wl in "encode:--source encode" "dub:--source dub" "encode_u8:--source encode --sample-type uint8" "dub_cc:--source dub --method ccoeff_normed" "snr0:--snr 0" "unrelated:--unrelated" "encode_whole:--source encode --exclusion whole" "dub_whole:--source dub --exclusion whole"; do
  name=${wl%%:*}; args=${wl#*:}
  timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --emulate-shards 0 --cpu-sample 128 $args > $O/bench_$name.json 2> $O/bench_$name.err || tail -3 $O/bench_$name.err
done
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/bench_*.json")):
    try:
        d=json.load(open(f)); r=d["roofline"]; g=r.get("diagnostics") or {}; p=d["parity"]
        print(f.split("/")[-1], round(d["value"]), round(d["ms_per_step"],3), {k: round(v,2) for k,v in (r.get("stage_ms") or {}).items()}, "pairs", g.get("pairs_transformed"), "band", g.get("band"), g.get("band_votes"), "susp", g.get("suspended"), "flagged", g.get("flagged"), "allpos", g.get("all_positions"), "ratios", round(g.get("max_bound_ratio"),3), round(g.get("max_bound_ratio_noncandidate"),3), "slb", round(g.get("max_slb_ratio_excluded"),3), "oracle", p["oracle_sample_searches"], "idx_err", p.get("max_idx_err_vs_oracle_sample"), "planted", d["config"].get("events_with_a_planted_answer"), p.get("max_shift_err_samples_vs_planted"), p.get("events_beyond_one_sample_of_planted"))
    except Exception as e: print(f, "ERR", e)
PY
