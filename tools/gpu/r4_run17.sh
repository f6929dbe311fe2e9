cd $GRAFT_REPO_ROOT
bash tools/gpu/ab_env.sh "LCGS_JOIN=0" "LCGS_JOIN=1" "LCGS_JOIN=2" 2>&1 | tee gpurun_out/r4_ab_join.log
