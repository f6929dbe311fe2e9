cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_gpu_fused.py tests/test_gpu_stages.py -x -q 2>&1 | tail -3
B="python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-train-step --no-stage-path --no-backward"
for rep in 1 2; do for v in "$@"; do
cp gpurun_in/liblcgs_$v.so luisacomputegaussiansplatting_amd/liblcgs_hip.so
timeout 100 $B 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$v', d['value'], 'batch', d['camera_batch']['value'], d['stages_ms'])"
done; done
