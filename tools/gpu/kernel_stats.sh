# Per-kernel statistics including the stage-level operators: gpurun -- bash tools/gpu/kernel_stats.sh
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/st
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/st/raw -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-train-step --no-backward --no-batch > gpurun_out/st/bench.json 2> gpurun_out/st/err.log
python3 profiles/summarize.py gpurun_out/st/raw > gpurun_out/st/stats.txt 2>&1
rm -rf gpurun_out/st/raw
cat gpurun_out/st/stats.txt
