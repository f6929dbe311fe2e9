cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
cp luisacomputegaussiansplatting_amd/liblcgs_hip.so /tmp/keep.so
for rep in 1 2; do for v in f32only inline fixup; do
cp gpurun_in/liblcgs_$v.so luisacomputegaussiansplatting_amd/liblcgs_hip.so
timeout 200 python bench.py --steps 60 --warmup 8 --no-cpu-baseline --no-stage-path --no-batch 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); fb=d['fwd_bwd']; ts=d.get('train_step',{}); fo=d.get('file_order',{})
print('$v fwd_bwd', fb['value'], fb['backward_stages_ms'], 'compact', fb.get('compact_rows',{}).get('value'), 'moving', fb.get('moving_camera',{}).get('value'), 'fit4', fb.get('multi_view_step_4',{}).get('lcgs_fit_views',{}).get('value'), 'file_order', fo.get('fwd_bwd',{}).get('value'), 'train', ts.get('visible_only',{}).get('value'), ts.get('visible_only_compact',{}).get('value'), ts.get('visible_only_fused',{}).get('value'))"
done; done 2>&1 | tee gpurun_out/r4_ab_f64b.log
cp /tmp/keep.so luisacomputegaussiansplatting_amd/liblcgs_hip.so
LCGS_SOAK=1500 LCGS_SOAK_REPORT=1 LCGS_SOAK_REPORT_FILE=gpurun_out/r4_soak_survey_fixup.json timeout -k 10 300 python -m pytest tests/test_gpu_soak.py -m gpu -q -s 2>&1 | grep "soak survey\] b\|passed\|failed\|\[soak\] kernel"
timeout -k 10 400 python -m pytest tests/test_gpu_backward.py tests/test_gpu_train.py tests/test_gpu_random_sweep.py tests/test_gpu_fullsize.py tests/test_gpu_comm.py -m gpu -q -x 2>&1 | tail -3
