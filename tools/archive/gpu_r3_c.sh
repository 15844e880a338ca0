# Round 3, third GPU call: what the box gives the host side (cgroup quota, scaling of the oracle leg), and whether VALU work
# and LDS traffic of different waves overlap on a CU (the question behind ifft_kernel's VALU-time + LDS-time = kernel time).
set -x
O=gpurun_out/r03c
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
(cat /sys/fs/cgroup/cpu.max; cat /sys/fs/cgroup/cpu.stat; cat /proc/loadavg; nproc; python -c "import os; print(len(os.sched_getaffinity(0)))"; cat /sys/kernel/mm/transparent_hugepage/enabled; lscpu | head -20; free -g) > $O/host.log 2>&1; cat $O/host.log
timeout 60 tools/ubench/valu_lds_overlap > $O/valu_lds_overlap.log 2>&1; cat $O/valu_lds_overlap.log
timeout 400 python tools/cpu_scaling.py > $O/cpu_scaling.log 2>&1; cat $O/cpu_scaling.log
