# A/B of tuning hooks (environment variables) on the exact stage path, one library, one box:
#   gpurun -- bash tools/gpu/ab_env_stage.sh "LCGS_STAGE_SIDE_COPY=0" "LCGS_STAGE_SIDE_COPY=1" ...
cd $GRAFT_REPO_ROOT
B="python bench.py --steps 80 --warmup 10 --no-cpu-baseline --no-train-step --no-backward --no-batch --no-spatial --no-moving-camera"
for rep in 1 2 3; do for v in "$@"; do
env $v timeout 200 $B 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); s=d['stage_path']; print('$v', 'fused', d['value'], 'stage exact', s['value'], 'deferred', s['deferred']['value'], 'diff', s['max_abs_diff_vs_fused'])"
done; done
