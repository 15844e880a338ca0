# PMC traffic (FETCH_SIZE / WRITE_SIZE passes) of the two secondary bench workloads, for roofline.traffic
set -x
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
pass() {  # tag, counters..., then the bench arguments after --
  tag=$1; shift; ctr=""
  while [ "$1" != "--" ]; do ctr="$ctr $1"; shift; done; shift
  rm -rf gpurun_out/prof_$tag
  rocprofv3 --pmc $ctr --output-format csv -d gpurun_out/prof_$tag -o $tag -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > gpurun_out/$tag.log 2>&1
}
pass u8_fetch FETCH_SIZE -- --sample-type uint8
pass u8_write WRITE_SIZE -- --sample-type uint8
pass c3_fetch FETCH_SIZE -- --window 120 --minutes 120 --events 375
pass c3_write WRITE_SIZE -- --window 120 --minutes 120 --events 375
python tools/summarize_pmc.py gpurun_out/pmc_u8.csv gpurun_out/prof_u8_fetch/u8_fetch_counter_collection.csv gpurun_out/prof_u8_write/u8_write_counter_collection.csv
python tools/summarize_pmc.py gpurun_out/pmc_c3.csv gpurun_out/prof_c3_fetch/c3_fetch_counter_collection.csv gpurun_out/prof_c3_write/c3_write_counter_collection.csv
grep "ifft\|mac_kernel\|^kernel" gpurun_out/pmc_u8.csv gpurun_out/pmc_c3.csv
