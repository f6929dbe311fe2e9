cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 700 python -m pytest tests -m gpu -q > gpurun_out/r4_t2.log 2>&1; echo "pytest rc=$?"; tail -25 gpurun_out/r4_t2.log
LCGS_SOAK=1500 LCGS_SOAK_REPORT=1 timeout -k 10 400 python -m pytest tests/test_gpu_soak.py -m gpu -q -s 2>&1 | grep -v "^\[parity" > gpurun_out/r4_soak_survey.log; echo "soak rc=$?"; grep "soak" gpurun_out/r4_soak_survey.log | cut -c1-1500
