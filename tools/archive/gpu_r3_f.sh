# ifft_kernel ablations (dev builds -DSUSHI_DEV_IFFT_ABL=n, tools/experiments): where its 14 ms go
set -x
O=gpurun_out/r03f
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export SUSHI_BENCH_CACHE=/tmp/sushi_bench_cache
timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --cpu-sample 2 > $O/warm.json 2> $O/warm.err; head -c 200 $O/warm.json; echo
for v in product abl9 abl1 abl2 abl3 abl4 abl5 abl6 abl7 abl8 product; do
  if [ $v = product ]; then unset SUSHI_HIP_LIB; else export SUSHI_HIP_LIB=$PWD/sushi_amd/lib/libsushi_hip_$v.so; fi
  timeout 120 python tools/stage_times.py --steps 5 --tag $v 2>/dev/null | tail -1 | tee -a $O/abl.log
done
