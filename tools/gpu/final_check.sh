# the round's closing check on the GPU box: the whole `-m gpu` suite in ONE pytest process, then __graft_entry__.smoke()
# (a heartbeat keeps a long quiet stretch from being taken for a hang): gpurun --timeout 1190 -- bash tools/gpu/final_check.sh
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/final
( while true; do sleep 55; echo "running $(date +%T)"; done ) &
HB=$!
timeout -k 10 1150 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/final/gputest.log 2>&1
rc=$?
kill $HB
tail -5 gpurun_out/final/gputest.log
if [ $rc -eq 0 ]; then
    python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/final/smoke.log 2>&1
    rc=$?
    tail -2 gpurun_out/final/smoke.log
fi
exit $rc
