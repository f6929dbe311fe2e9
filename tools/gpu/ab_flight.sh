# A/B of the frames-in-flight hooks (environment variables) and of prebuilt libraries on one box:
#   gpurun -- bash tools/gpu/ab_flight.sh "LIB=base A=1 B=2" "LIB=other A=3" ...   (each argument: one space-separated setting;
#   LIB=<tag> first copies gpurun_in/liblcgs_<tag>.so over the library, default: the library as shipped)
# Prints, per setting and repetition: in-order fps, camera batch fps, fwd+bwd Msplats/s, lcgs_fit_views (4 views) Msplats/s.
cd $GRAFT_REPO_ROOT
cp luisacomputegaussiansplatting_amd/liblcgs_hip.so /tmp/lcgs_shipped.so
B="python bench.py --steps 60 --warmup 8 --no-cpu-baseline --no-train-step --no-stage-path --no-spatial"
for rep in 1 2; do for v in "$@"; do
lib=$(echo "$v" | grep -o 'LIB=[a-z0-9_]*' | cut -d= -f2)
if [ -n "$lib" ]; then cp gpurun_in/liblcgs_$lib.so luisacomputegaussiansplatting_amd/liblcgs_hip.so; else cp /tmp/lcgs_shipped.so luisacomputegaussiansplatting_amd/liblcgs_hip.so; fi
env $v timeout 300 $B 2>gpurun_out/ab_flight.err | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
fb=d.get('fwd_bwd',{})
print('$v | fwd', d['value'], '| batch', d.get('camera_batch',{}).get('value'), d.get('camera_batch',{}).get('images_equal'),
      '| fwd_bwd', fb.get('value'), '| fit4', fb.get('multi_view_step_4',{}).get('lcgs_fit_views',{}).get('value'),
      '1by1', fb.get('multi_view_step_4',{}).get('one_by_one',{}).get('value'), '| render ms', d['stages_ms'].get('render'), flush=True)" || { echo "$v FAILED"; tail -5 gpurun_out/ab_flight.err; }
done; done
cp /tmp/lcgs_shipped.so luisacomputegaussiansplatting_amd/liblcgs_hip.so
