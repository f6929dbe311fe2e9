# A/B of prebuilt libraries on the backward legs: gpurun -- bash tools/gpu/ab_bwd.sh <tagA> <tagB> ... (gpurun_in/liblcgs_<tag>.so)
cd $GRAFT_REPO_ROOT
cp luisacomputegaussiansplatting_amd/liblcgs_hip.so /tmp/keep.so
for rep in 1 2 3; do for v in "$@"; do
cp gpurun_in/liblcgs_$v.so luisacomputegaussiansplatting_amd/liblcgs_hip.so
timeout 200 python bench.py --steps 60 --warmup 8 --no-cpu-baseline --no-stage-path --no-spatial --no-batch --no-moving-camera 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); fb=d['fwd_bwd']; ts=d.get('train_step',{})
print('$v fwd', d['value'], 'fwd_bwd', fb['value'], fb['backward_stages_ms'], 'compact', fb.get('compact_rows',{}).get('value'), 'train fused', ts.get('visible_only_fused',{}).get('value'))"
done; done
cp /tmp/keep.so luisacomputegaussiansplatting_amd/liblcgs_hip.so
