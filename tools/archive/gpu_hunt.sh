set -x
O=gpurun_out/hunt
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 400 python tools/bound_hunt.py 36 1 > $O/bound_hunt.log 2>&1; tail -12 $O/bound_hunt.log
timeout 200 python tools/prop_hunt.py 600 7 > $O/prop_hunt.log 2>&1; tail -3 $O/prop_hunt.log
