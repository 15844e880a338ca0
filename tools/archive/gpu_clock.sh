# effective shader clock per kernel = GRBM_GUI_ACTIVE / dispatch duration (dev)
set -x
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline"
rm -rf gpurun_out/prof_clk
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/prof_clk -o clk -- $B > gpurun_out/clk.log 2>&1
ls gpurun_out/prof_clk
python - <<'PY'
import csv, collections
dur = {}
for r in csv.DictReader(open('gpurun_out/prof_clk/clk_kernel_trace.csv')):
    dur[r['Dispatch_Id']] = (r['Kernel_Name'][:50], int(r['End_Timestamp']) - int(r['Start_Timestamp']))
agg = collections.defaultdict(list)
for r in csv.DictReader(open('gpurun_out/prof_clk/clk_counter_collection.csv')):
    if r['Counter_Name'] == 'GRBM_GUI_ACTIVE' and r['Dispatch_Id'] in dur:
        name, ns = dur[r['Dispatch_Id']]
        agg[name].append((float(r['Counter_Value']), ns))
for k, v in agg.items():
    if 'ifft' in k or 'mac' in k:
        print('CLK', k, [round(c / ns, 3) for c, ns in v], 'GHz; ns', [ns for c, ns in v])
PY
