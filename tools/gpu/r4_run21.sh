# cull bound rows: the new tests, then A/B of LCGS_CULL_BOUND (forward stages + backward legs)
cd $GRAFT_REPO_ROOT
timeout -k 10 400 python -m pytest tests/test_gpu_cull_bound.py tests/test_gpu_ingest.py tests/test_gpu_fused.py -m gpu -q -x 2>&1 | tail -5 || exit 1
bash tools/gpu/ab_env.sh "LCGS_CULL_BOUND=0" "LCGS_CULL_BOUND=1" 2>&1 | tee gpurun_out/r4_ab_cullbound.log
bash tools/gpu/ab_env_bwd2.sh "LCGS_CULL_BOUND=0" "LCGS_CULL_BOUND=1" 2>&1 | tee -a gpurun_out/r4_ab_cullbound.log
