# Round-3 evidence at the packed-half-operand kernels (profiles/r03/): FETCH / WRITE passes first (so that the bench lines
# carry roofline.traffic), kernel trace, the bench lines, smoke; WITH_TESTS=1 adds the whole GPU suite, WITH_CFG4=1 the
# configs[4] line with its own counters.  Every command under its own timeout.
set -x
O=gpurun_out/r03final2
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export SUSHI_BENCH_CACHE=/tmp/sushi_bench_cache
B="python bench.py --steps 5 --warmup 1 --no-cpu-baseline --cpu-sample 2"
timeout 300 $B > $O/warm.json 2> $O/warm.err            # builds the stream cache
rm -rf gpurun_out/prof_*
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_kt -o kt -- $B > $O/kt.log 2>&1
timeout 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof_fetch -o fetch -- $B > $O/fetch.log 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d gpurun_out/prof_write -o write -- $B > $O/write.log 2>&1
cp gpurun_out/prof_kt/kt_kernel_stats.csv $O/cfg2_kernel_stats.csv
python tools/summarize_pmc.py $O/cfg2_pmc_summary.csv $(find gpurun_out/prof_fetch gpurun_out/prof_write -name '*counter_collection.csv')
python tools/make_pmc_traffic.py $O/cfg2_pmc_summary.csv profiles/pmc_traffic.json "config2/fft/float32/3000/w120/m120/n1" "$COMMIT" > /dev/null
if [ "$WITH_TCP" = 1 ]; then
timeout 200 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum --kernel-include-regex "ifft_kernel|mac_kernel" --output-format csv -d gpurun_out/prof_tcp -o tcp -- python tools/stage_times.py --steps 2 --tag tcp > $O/tcp.log 2>&1
python tools/summarize_pmc.py $O/cfg2_tcp_summary.csv $(find gpurun_out/prof_tcp -name '*counter_collection.csv'); cat $O/cfg2_tcp_summary.csv
fi
cp profiles/pmc_traffic.json $O/pmc_traffic.json
timeout 400 python bench.py --steps 20 --warmup 5 > $O/bench_cfg2_n1.json 2> $O/bench_cfg2_n1.err; head -c 300 $O/bench_cfg2_n1.json; echo
timeout 300 python bench.py --steps 10 --warmup 3 --hard-frac 0.05 --no-cpu-baseline > $O/bench_cfg2_hard_n1.json 2> $O/e2.err
timeout 300 python bench.py --steps 10 --warmup 3 --sample-type uint8 --no-cpu-baseline > $O/bench_cfg2_u8_n1.json 2> $O/e1.err
timeout 300 python bench.py --steps 10 --warmup 3 --method ccoeff_normed --no-cpu-baseline > $O/bench_cfg2_ccoeff_n1.json 2> $O/e3.err
timeout 300 python bench.py --config 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_cfg1_n1.json 2> $O/e4.err
if [ "$WITH_CFG4" = 1 ]; then
timeout 600 python bench.py --config 4 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_cfg4_n1.json 2> $O/e5.err
fi
if [ "$WITH_TESTS" = 1 ]; then
timeout 900 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
fi
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $O/smoke.log
timeout 60 python tools/kernel_resources.py sushi_fft > $O/kernel_resources_fft.txt 2>&1
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r03final2/bench_*.json")):
    try:
        d=json.load(open(f)); r=d["roofline"]; p=d["parity"]
        print(f.split("/")[-1], round(d["value"]), round(d["ms_per_step"],2), r.get("stage_ms"), "frac", round(r["frac"],3), "traffic", r.get("traffic"), "oracle", p["oracle_sample_searches"], "idx_err", p.get("max_idx_err_vs_oracle_sample"), "cpu", d["cpu_baseline"] and round(d["cpu_baseline"]["value"],1), r.get("diagnostics"))
    except Exception as e: print(f, "ERR", e)
PY
