# A/B of prebuilt libraries, forward only with stage times: gpurun -- bash tools/gpu/ab_fwd_libs.sh <tagA> <tagB> ... (gpurun_in/liblcgs_<tag>.so)
cd $GRAFT_REPO_ROOT
B="python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-train-step --no-stage-path --no-spatial --no-backward --no-batch"
for rep in 1 2 3; do for v in "$@"; do
cp gpurun_in/liblcgs_$v.so luisacomputegaussiansplatting_amd/liblcgs_hip.so
timeout 200 $B 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$v', d['value'], 'moving', d.get('moving_camera',{}).get('value'), d['stages_ms'], flush=True)"
done; done
