set -x
O=gpurun_out/r03c
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 60 tools/ubench/valu_lds_overlap > $O/valu_lds_overlap.log 2>&1; cat $O/valu_lds_overlap.log
