cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
bash tools/gpu/ab_flight.sh "X=0" "LCGS_BATCH_IN_FLIGHT=3" "LCGS_BATCH_IN_FLIGHT=3 GPU_MAX_HW_QUEUES=8" "LCGS_BATCH_IN_FLIGHT=4 GPU_MAX_HW_QUEUES=8" \
  "LCGS_BATCH_IN_FLIGHT=3 GPU_MAX_HW_QUEUES=8 LCGS_RENDER_WGS_IN_FLIGHT=6" "LCGS_AUX_PRIORITY=same" "LCGS_AUX_PRIORITY=same LCGS_BATCH_IN_FLIGHT=3 GPU_MAX_HW_QUEUES=8" \
  "LCGS_BWD_WGS_IN_FLIGHT=5" "LCGS_BWD_WGS_IN_FLIGHT=4" "GPU_MAX_HW_QUEUES=8" 2>&1 | tee gpurun_out/r4_ab_flight2.log
cp luisacomputegaussiansplatting_amd/liblcgs_hip.so /tmp/keep.so
bash tools/gpu/ab_compare.sh base bw7 nonr 2>&1 | tee gpurun_out/r4_ab_bw.log
for v in base nonr; do
cp gpurun_in/liblcgs_$v.so luisacomputegaussiansplatting_amd/liblcgs_hip.so
LCGS_SOAK=1500 LCGS_SOAK_REPORT=1 LCGS_SOAK_REPORT_FILE=gpurun_out/r4_soak_survey_$v.json timeout -k 10 300 python -m pytest tests/test_gpu_soak.py -m gpu -q -s 2>&1 | grep "soak survey\] b\|passed\|failed"
done
cp /tmp/keep.so luisacomputegaussiansplatting_amd/liblcgs_hip.so
