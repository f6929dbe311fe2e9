# The judged artefacts in one call (late round 4): full GPU suite, 1500-draw soak, profile round, timelines, N > 1 rehearsal.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/final
timeout -k 10 600 python -m pytest tests -m gpu -q > gpurun_out/final/gputest.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/final/gputest.log
bash tools/gpu/soak.sh 1500 > gpurun_out/final/soak_tail.log 2>&1; echo "soak rc=$?"; tail -2 gpurun_out/final/soak_tail.log
bash tools/profile_round.sh > gpurun_out/final/profile_round.log 2>&1; echo "profile_round rc=$?"; tail -3 gpurun_out/final/profile_round.log
bash tools/gpu/timeline_forward.sh > /dev/null 2>&1; cp gpurun_out/tl/timeline.txt gpurun_out/final/timeline_forward.txt
bash tools/gpu/timeline_step.sh > /dev/null 2>&1; cp gpurun_out/tl2/timeline.txt gpurun_out/final/timeline_step.txt
LCGS_BENCH_FORCE_DIST=1 timeout 300 python bench.py --no-cpu-baseline > gpurun_out/final/bench_force_dist.json 2> gpurun_out/final/bench_force_dist.err; echo "force_dist rc=$?"
