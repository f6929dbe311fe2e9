cd $GRAFT_REPO_ROOT
bash tools/gpu/ab_env_bwd2.sh "X=1" 2>&1 | tee gpurun_out/r4_ab_stride.log
timeout -k 10 500 python -m pytest tests/test_gpu_backward.py tests/test_gpu_train.py tests/test_gpu_random_sweep.py tests/test_gpu_fullsize.py tests/test_gpu_owner.py tests/test_gpu_pathological.py -m gpu -q -x 2>&1 | tail -3
