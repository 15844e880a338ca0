# every search of the job against the oracle (the CPU leg's whole 25 s budget), workload by workload
O=gpurun_out/r06x
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export SUSHI_BENCH_CACHE=/tmp/sushi_bench_cache
line() {   # name args...
  name=$1; shift 1
  timeout 900 python bench.py --steps 10 --warmup 3 --emulate-shards 0 --cpu-sample 3000 "$@" > $O/full_oracle_$name.json 2> $O/b.err
  python - <<PY | tee -a $O/sweep.txt
import json
try:
    d=json.load(open("$O/full_oracle_$name.json")); r=d["roofline"]; g=r.get("diagnostics") or {}; p=d["parity"]
    print("$name", round(d["ms_per_step"],3), "pairs", g.get("pairs_transformed"), "oracle searches", p["oracle_sample_searches"], "max idx err", p.get("max_idx_err_vs_oracle_sample"), "max score err", p.get("max_abs_score_err_vs_oracle_sample"), "flagged", g.get("flagged"), "cpu", d["cpu_baseline"] and round(d["cpu_baseline"]["value"],1))
except Exception as e:
    print("$name", "FAILED", e, open("$O/b.err").read()[-800:])
PY
}
line dub --source dub
line partial --source partial
line encode --source encode
line snr6 --snr 6
line snr0 --snr 0
line unrelated --unrelated
line hard --hard-frac 0.05
line cc --method ccoeff_normed
line u8 --sample-type uint8
line dub_cc --source dub --method ccoeff_normed
line dub_u8 --source dub --sample-type uint8
