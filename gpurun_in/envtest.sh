cd $GRAFT_REPO_ROOT
B="python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-train-step --no-stage-path --no-backward --no-batch"
for rep in 1 2; do
for e in "X=1" "HIP_FORCE_DEV_KERNARG=1" "HIP_FORCE_DEV_KERNARG=0"; do
env $e timeout 200 $B 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$e', d['value'], d['stages_ms'])"
done; done
