# A/B of environment hooks on the forward legs incl. the camera batch: gpurun -- bash tools/gpu/ab_env_batch.sh "A=1" "B=2" ...
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do for v in "$@"; do
env $v timeout 200 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-train-step --no-stage-path --no-spatial --no-backward --no-moving-camera 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$v fwd', d['value'], 'batch', d['camera_batch']['value'], d['camera_batch']['images_equal'], flush=True)"
done; done
