#!/bin/bash
# copy a tools/gpu_final_r5.sh run (gpurun_out/<dir>) into the tracked profiles/r05/final/ and profiles/pmc_traffic.json
set -e
D=gpurun_out/${1:?dir}
mkdir -p profiles/r05/final
for f in $D/*_kernel_stats.csv $D/*_pmc_summary.csv $D/bench_*_n1.json $D/pytest_gpu.log $D/smoke.log $D/kernel_resources_*.txt $D/latency.json $D/pmc_traffic.json; do cp $f profiles/r05/final/; done
cp $D/pmc_traffic.json profiles/pmc_traffic.json
sed -i "s#$D/#profiles/r05/final/#g" profiles/pmc_traffic.json profiles/r05/final/pmc_traffic.json
