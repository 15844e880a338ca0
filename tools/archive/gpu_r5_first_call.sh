# Next round's first GPU call (about 6 GPU-minutes; drop the pytest block to save 2.5): where do the 12.8 ms of the store-wave variant's compute waves go?
# Build the libraries first, here: tools/experiments/build_r5_first_call_variants.sh.  Then
#   gpurun --timeout 600 -- 'bash tools/gpu_r5_first_call.sh'
# CORRECT results: product, la2, bound2, bound2_la2 (the two prepared gains: adopt what holds, then tools/gpu_final_r4.sh), sw, iso_w5, iso_w5_2cu
#   (bench.py: parity sample + stage times; bound2: also tests/test_pair_exclusion.py + tests/test_gpu_parity.py under it)
# GARBAGE Y (timing only; built with r05_timing_only_on_top.patch, which skips everything behind mac_kernel: stage_times.py --steps 5): no_y, iso_no_wait, iso_no_wait_no_store, iso_no_consumer
#   iso_w5                 product + an idle fifth wave                        -> what 320-thread workgroups cost by themselves
#   iso_w5_2cu             ... + the queue's 30 KB of LDS                      -> ... at two workgroups per CU
#   iso_no_consumer        queue pushes, no reserve wait, store wave returns   -> the pushes alone (compare with no_y: 1.7 ms)
#   iso_no_wait_no_store   ... store wave polls and frees the slots, no stores -> the store wave's presence
#   iso_no_wait            ... and stores                                      -> the stores from a fifth wave
#   sw                     the whole design                                    -> + the reserve wait
#   sw2                    ... two (4 + 1)-wave groups per workgroup             -> is it the PLACEMENT of five-wave workgroups?  At 164 VGPRs a SIMD
#                          holds three waves; two five-wave workgroups fit a CU only if they double up on different SIMDs, and the
#                          counters of the one-group build say 7 resident waves per CU on average, not 10 (SQ_WAVE_CYCLES over the kernel's time)
#                          MEASURED at the end of round 4: 11.6 ms against 16.1.  Its knobs: sw2_q8 (eight slots), sw2_q8_r4 (+ rounds of four
#                          outputs in classes 6 and 12: a producer may then run ahead of the store wave), sw2_m3 (three slots drained per step),
#                          sw2_no_wait (garbage: what the reserve wait costs now)
set -x
O=gpurun_out/${OUT:-r5_first}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export SUSHI_BENCH_CACHE=/tmp/sushi_bench_cache
for V in product la2 bound2 bound2_la2 sw2 sw2_q8 sw2_q8_r4 sw2_m3 sw iso_w5 iso_w5_2cu; do
  if [ $V = product ]; then unset SUSHI_HIP_LIB; else export SUSHI_HIP_LIB=$PWD/sushi_amd/lib/libsushi_hip_$V.so; fi
  [ $V = product ] || [ -f "$SUSHI_HIP_LIB" ] || { echo "$V: not built" | tee -a $O/notes.txt; continue; }
  timeout 60 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --cpu-sample 64 > $O/bench_$V.json 2> $O/b_$V.err; tail -n 2 $O/b_$V.err
done
if [ -f $PWD/sushi_amd/lib/libsushi_hip_bound2_la2.so ]; then
  SUSHI_HIP_LIB=$PWD/sushi_amd/lib/libsushi_hip_bound2_la2.so timeout 400 python -m pytest tests/test_pair_exclusion.py tests/test_gpu_parity.py tests/test_ccoeff.py -m gpu -q -x > $O/pytest_bound2_la2.log 2>&1; tail -n 4 $O/pytest_bound2_la2.log
fi
for V in no_y sw2_no_wait iso_no_consumer iso_no_wait_no_store iso_no_wait; do
  export SUSHI_HIP_LIB=$PWD/sushi_amd/lib/libsushi_hip_$V.so
  [ -f "$SUSHI_HIP_LIB" ] || { echo "$V: not built" | tee -a $O/notes.txt; continue; }
  timeout 60 python tools/stage_times.py --steps 5 --tag $V 2>$O/st_$V.err | tail -n 1 | tee -a $O/garbage_timing.jsonl
done
# per-wave timing records of one launch (tools/read_mac_probe.py): where a producer wave's time goes (probe_no_y: garbage Y, one 13 s run)
for V in probe_product probe_sw probe_no_y; do
  export SUSHI_HIP_LIB=$PWD/sushi_amd/lib/libsushi_hip_$V.so
  [ -f "$SUSHI_HIP_LIB" ] || continue
  timeout 60 python tools/read_mac_probe.py --tag $V 2>$O/probe_$V.err | tail -n 1 | tee -a $O/mac_probe.jsonl
done
unset SUSHI_HIP_LIB
python - <<PY
import json
for V in "product la2 bound2 bound2_la2 sw2 sw2_q8 sw2_q8_r4 sw2_m3 sw iso_w5 iso_w5_2cu".split():
    try:
        d=json.load(open("$O/bench_%s.json" % V)); r=d["roofline"]; p=d["parity"]
        print(V, round(d["value"]), round(d["ms_per_step"],2), {k: round(v,3) for k,v in r["stage_ms"].items()}, r["diagnostics"]["flagged"], p["oracle_sample_searches"], p.get("max_idx_err_vs_oracle_sample"))
    except Exception as e: print(V, "ERR", e)
PY
cat $O/garbage_timing.jsonl
