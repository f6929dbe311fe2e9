# A/B of tuning hooks (environment variables) on the forward+backward step, one library, one box:
#   gpurun -- bash tools/gpu/ab_env_bwd.sh "LCGS_X=0" "LCGS_X=1" ...
cd $GRAFT_REPO_ROOT
B="python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-train-step --no-stage-path --no-spatial --no-batch --no-moving-camera"
for rep in 1 2; do for v in "$@"; do
env $v timeout 200 $B 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); f=d['fwd_bwd']; print('$v', 'fwd', d['value'], 'fwd_bwd', f['value'], f['ms_per_step'], 'compact', f['compact_rows']['value'])"
done; done
