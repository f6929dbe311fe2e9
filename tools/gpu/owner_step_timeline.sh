# The ownership step (N = 1 through RCCL, no read-back) as a timeline: gpurun -- bash tools/gpu/owner_step_timeline.sh
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/tlo
python3 tools/gpu/owner_step_driver.py > gpurun_out/tlo/plain.txt 2> gpurun_out/tlo/plain.err
LCGS_DRIVER_ASYNC=0 python3 tools/gpu/owner_step_driver.py > gpurun_out/tlo/plain_sync.txt 2>> gpurun_out/tlo/plain.err
rocprofv3 --kernel-trace --hip-trace --marker-trace --output-format csv -d gpurun_out/tlo/raw -- python3 tools/gpu/owner_step_driver.py > gpurun_out/tlo/driver.txt 2> gpurun_out/tlo/err.log
python3 tools/gpu/owner_step_timeline.py gpurun_out/tlo/raw > gpurun_out/tlo/timeline.txt 2>&1
rm -rf gpurun_out/tlo/raw
cat gpurun_out/tlo/plain.txt gpurun_out/tlo/plain_sync.txt gpurun_out/tlo/timeline.txt
