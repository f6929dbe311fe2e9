cd $GRAFT_REPO_ROOT
timeout -k 10 400 python -m pytest tests/test_gpu_stages.py tests/test_gpu_pathological.py -m gpu -q -x 2>&1 | tail -3 || exit 1
for rep in 1 2 3; do
timeout 200 python bench.py --steps 60 --warmup 8 --no-cpu-baseline --no-train-step --no-backward --no-batch --no-spatial --no-moving-camera 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('stage_path', d['stage_path']['value'], d['stage_path']['ms_per_step'], 'deferred', d['stage_path']['deferred']['value'], 'fwd', d['value'], flush=True)"
done
