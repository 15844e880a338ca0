cd $GRAFT_REPO_ROOT/tools/ubench && ./valu_rate
