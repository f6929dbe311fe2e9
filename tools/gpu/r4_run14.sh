cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_owner.py -m gpu -q -x 2>&1 | tail -20
