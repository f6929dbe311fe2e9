# A/B of environment hooks on the backward legs: gpurun -- bash tools/gpu/ab_env_bwd.sh "A=1" "B=2" ...
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do for v in "$@"; do
env $v timeout 200 python bench.py --steps 60 --warmup 8 --no-cpu-baseline --no-stage-path --no-batch 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); fb=d['fwd_bwd']; ts=d.get('train_step',{}); fo=d.get('file_order',{})
print('$v fwd', d['value'], 'fwd_bwd', fb['value'], fb['ms_per_step'], 'compact', fb.get('compact_rows',{}).get('value'), 'moving', fb.get('moving_camera',{}).get('value'), 'fit4', fb.get('multi_view_step_4',{}).get('lcgs_fit_views',{}).get('value'), 'file_order', fo.get('fwd_bwd',{}).get('value'), 'train', ts.get('dense',{}).get('value'), ts.get('visible_only',{}).get('value'), flush=True)"
done; done
