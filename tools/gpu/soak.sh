# opt-in soak (tests/test_gpu_soak.py) with a heartbeat so that a long quiet run is not taken for a hang: bash tools/gpu/soak.sh <draws>
cd $GRAFT_REPO_ROOT
( while true; do sleep 50; echo "soak running $(date +%T)"; done ) &
HB=$!
LCGS_SOAK=${1:-200} timeout -k 10 1000 python -m pytest tests/test_gpu_soak.py -m gpu -q -x > gpurun_out/soak.log 2>&1
rc=$?
kill $HB
tail -5 gpurun_out/soak.log
exit $rc
