# Round 4: A/B of variant libraries: stage times of each (VARIANTS="product name ..."), then the kernel-level parity tests under TESTLIB
set -x
O=gpurun_out/${OUT:-r4ab}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export SUSHI_BENCH_CACHE=/tmp/sushi_bench_cache
timeout 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --cpu-sample 64 > $O/bench_product.json 2> $O/b.err; tail -2 $O/b.err   # (writes the stream cache)
for v in $VARIANTS; do
  if [ $v = product ]; then unset SUSHI_HIP_LIB; else export SUSHI_HIP_LIB=$PWD/sushi_amd/lib/libsushi_hip_$v.so; fi
  timeout 150 python tools/stage_times.py --steps 10 --tag $v 2>$O/st_$v.err | tail -1 | tee -a $O/ab.log
done
unset SUSHI_HIP_LIB
if [ -n "$TESTLIB" ]; then
  export SUSHI_HIP_LIB=$PWD/sushi_amd/lib/libsushi_hip_$TESTLIB.so
  timeout 600 python -m pytest ${TESTS:-tests/test_gpu_parity.py tests/test_ccoeff.py} -m gpu -q -x > $O/pytest_$TESTLIB.log 2>&1; tail -5 $O/pytest_$TESTLIB.log
fi
