# A/B of tuning hooks (environment variables) with one library on one box:
#   gpurun -- bash tools/gpu/ab_env.sh "LCGS_DEPTH_BUCKETS=0" "LCGS_DEPTH_BUCKETS=1024" ...
# Each setting runs the short forward bench twice, interleaved (clock / placement drift shows up as spread).
cd $GRAFT_REPO_ROOT
B="python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-train-step --no-stage-path --no-spatial --no-backward --no-batch"
for rep in 1 2; do for v in "$@"; do
env $v timeout 200 $B 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$v', d['value'], 'moving', d.get('moving_camera',{}).get('value'), d['stages_ms'])"
done; done
