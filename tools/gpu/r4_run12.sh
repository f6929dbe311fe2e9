cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_owner.py -m gpu -q 2>&1 | tail -30
timeout -k 10 900 python -m pytest tests/test_gpu_comm.py -m gpu -q -x -k "bench" 2>&1 | tail -30
