# A/B of variant libraries sushi_amd/lib/libsushi_hip_<v>.so (VARIANTS="a b c") against the product on one box: stage times only
set -x
O=gpurun_out/vars2
mkdir -p $O; rm -f $O/ab.log
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export SUSHI_BENCH_CACHE=/tmp/sushi_bench_cache
timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --cpu-sample 2 > $O/warm.json 2> $O/b.err; tail -1 $O/b.err
for rep in 1 2; do
for v in product $VARIANTS; do
  if [ $v = product ]; then unset SUSHI_HIP_LIB; else export SUSHI_HIP_LIB=$PWD/sushi_amd/lib/libsushi_hip_$v.so; fi
  timeout 120 python tools/stage_times.py --steps 10 --tag $v 2>/dev/null | tail -1 | tee -a $O/ab.log
done
done
