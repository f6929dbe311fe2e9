# A/B of prebuilt libraries on one box: gpurun -- bash tools/gpu/ab_compare.sh <tagA> <tagB> (expects gpurun_in/liblcgs_<tag>.so)
cd $GRAFT_REPO_ROOT

B="python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-train-step --no-stage-path"
for rep in 1 2; do for v in "$@"; do
cp gpurun_in/liblcgs_$v.so luisacomputegaussiansplatting_amd/liblcgs_hip.so
timeout 100 $B 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$v', d['value'], 'batch', d['camera_batch']['value'], 'fwdbwd', d['fwd_bwd']['value'], d['stages_ms'])"
done; done
