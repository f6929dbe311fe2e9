cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
hipcc -O3 --offload-arch=gfx950 tools/microbench/cu_mask_probe.hip -o /tmp/cu_mask_probe 2>/dev/null && /tmp/cu_mask_probe > gpurun_out/r4_cu_mask_probe.txt 2>&1
cat gpurun_out/r4_cu_mask_probe.txt
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r4_t1.log 2>&1; echo "pytest rc=$?"; tail -15 gpurun_out/r4_t1.log
