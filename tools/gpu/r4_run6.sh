cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
rocprofv3 -L 2>/dev/null | grep -o "SQ_ACTIVE_INST_[A-Z_]*\|SQ_WAIT_INST_ANY\|SQ_WAIT_ANY\|SQ_BUSY_CYCLES\|SQ_INST_CYCLES_[A-Z_]*" | sort -u | tr '\n' ' '; echo
timeout -k 10 700 python -m pytest tests -m gpu -q -x > gpurun_out/r4_t6.log 2>&1; echo "pytest rc=$?"; tail -8 gpurun_out/r4_t6.log
python bench.py --no-cpu-baseline > gpurun_out/r4_b6.json 2> gpurun_out/r4_b6.err; echo "bench rc=$?"; python - <<'P'
import json
d=json.loads(open('gpurun_out/r4_b6.json').readline())
print(d['value'], d['per_frame_events'], d['roofline'].get('profile_errors'), d['frame_roofline'].get('target_60pct_hbm'))
print(d['fwd_bwd'].get('roofline'))
P
