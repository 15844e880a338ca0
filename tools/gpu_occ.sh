# mean waves per CU of the FFT path's kernels (dev): default build vs the LDS-padded one
set -x
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline"
for lib in sushi_amd/lib/libsushi_hip.so gpurun_scratch/libsushi_pad.so; do
tag=$(basename $lib .so)
SUSHI_HIP_LIB=$lib rocprofv3 --pmc MeanOccupancyPerCU MeanOccupancyPerActiveCU --output-format csv -d gpurun_out/prof_occ_$tag -o occ -- $B > gpurun_out/occ_$tag.log 2>&1
python - <<PY
import csv, collections
agg = collections.defaultdict(list)
for r in csv.DictReader(open('gpurun_out/prof_occ_$tag/occ_counter_collection.csv')):
    agg[(r['Kernel_Name'][:60], r['Counter_Name'])].append(float(r['Counter_Value']))
for k, v in sorted(agg.items()):
    if 'ifft' in k[0] or 'mac' in k[0]:
        print('OCC', '$tag', k, sum(v) / len(v))
PY
done
