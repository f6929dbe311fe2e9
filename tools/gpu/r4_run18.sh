cd $GRAFT_REPO_ROOT
bash tools/gpu/ab_bwd.sh base g2dr 2>&1 | tee gpurun_out/r4_ab_g2dr.log
timeout -k 10 400 python -m pytest tests/test_gpu_backward.py tests/test_gpu_train.py tests/test_gpu_random_sweep.py tests/test_gpu_fullsize.py tests/test_gpu_owner.py -m gpu -q -x 2>&1 | tail -3
