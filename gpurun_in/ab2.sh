cd $GRAFT_REPO_ROOT
B="python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-batch --no-train-step --no-stage-path --no-backward"
for v in "$@"; do
cp gpurun_in/liblcgs_$v.so luisacomputegaussiansplatting_amd/liblcgs_hip.so
timeout 120 $B 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$v', d['value'], d['stages_ms'])"
done
