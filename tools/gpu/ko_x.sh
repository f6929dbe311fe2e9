cd $GRAFT_REPO_ROOT
B="python bench.py --steps 200 --warmup 30 --no-cpu-baseline --no-train-step --no-stage-path --no-spatial --no-backward --no-batch --no-moving-camera"
for rep in 1 2 3; do for v in "$@"; do
env $v timeout 200 $B 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$v', d['value'], d['per_frame_events']['median_ms'], flush=True)"
done; done
