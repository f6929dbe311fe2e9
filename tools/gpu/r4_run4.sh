cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/tlb
python3 tools/gpu/batch_driver.py 40 batch
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tlb/raw -- python3 tools/gpu/batch_driver.py 40 batch > gpurun_out/tlb/out.txt 2> gpurun_out/tlb/err.log
cat gpurun_out/tlb/out.txt
python3 profiles/timeline_window.py gpurun_out/tlb/raw 4 > gpurun_out/r4_timeline_batch.txt 2>&1
rm -rf gpurun_out/tlb/raw
cat gpurun_out/r4_timeline_batch.txt
