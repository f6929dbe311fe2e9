cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
bash tools/gpu/ab_bwd.sh base striplast 2>&1 | tee gpurun_out/r4_ab_striplast.log
timeout -k 10 300 python -m pytest tests/test_gpu_backward.py tests/test_gpu_train.py tests/test_gpu_random_sweep.py tests/test_gpu_fullsize.py -m gpu -q -x 2>&1 | tail -3
