#!/bin/bash
# copy a tools/gpu_final_r6.sh run (gpurun_out/<dir>) into the tracked profiles/<round>/final/ and profiles/pmc_traffic.json
# usage: install_evidence.sh <dir under gpurun_out> [round dir, default r06]
set -e
D=gpurun_out/${1:?dir}
R=${2:-r06}
mkdir -p profiles/$R/final
for f in $D/*_kernel_stats.csv $D/*_pmc_summary.csv $D/bench_*_n1.json $D/pytest_gpu.log $D/smoke.log $D/kernel_resources_*.txt $D/latency.json $D/call_breakdown.json $D/excluded_audit_hunt.jsonl $D/bound_hunt.txt $D/pmc_traffic.json; do [ -f $f ] && cp $f profiles/$R/final/; done
cp $D/pmc_traffic.json profiles/pmc_traffic.json
sed -i "s#$D/#profiles/$R/final/#g" profiles/pmc_traffic.json profiles/$R/final/pmc_traffic.json
