cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_owner.py tests/test_gpu_comm.py -m gpu -q -x 2>&1 | tail -5
