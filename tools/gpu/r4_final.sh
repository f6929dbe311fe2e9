# The judged artefacts of round 4 in one call: tests, profile round, timelines, rehearsal of the N > 1 legs.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/final
timeout -k 10 600 python -m pytest tests -m gpu -q > gpurun_out/final/gputest.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/final/gputest.log
bash tools/profile_round.sh > gpurun_out/final/profile_round.log 2>&1; echo "profile_round rc=$?"; tail -3 gpurun_out/final/profile_round.log
bash tools/gpu/timeline_forward.sh > /dev/null 2>&1; cp gpurun_out/tl/timeline.txt gpurun_out/final/timeline_forward.txt
bash tools/gpu/timeline_step.sh > /dev/null 2>&1; cp gpurun_out/tl2/timeline.txt gpurun_out/final/timeline_step.txt
LCGS_BENCH_FORCE_DIST=1 timeout 300 python bench.py --no-cpu-baseline > gpurun_out/final/bench_force_dist.json 2> gpurun_out/final/bench_force_dist.err; echo "force_dist rc=$?"
python - <<'P'
import json
d=json.loads(open('gpurun_out/prof/bench_default.json').readline())
print(d['value'], d['per_frame_events']['median_ms'], d['roofline']['frac'], d['roofline'].get('valu_issue',{}) and d['roofline']['valu_issue'].get('frac'), d['fwd_bwd']['value'], d['camera_batch']['value'], d['parity'])
P
