# One forward+backward step as a kernel timeline: gpurun -- bash tools/gpu/timeline_step.sh
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/tl2
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl2/raw -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-train-step --no-stage-path --no-batch > gpurun_out/tl2/bench.json 2> gpurun_out/tl2/err.log
python3 profiles/timeline_step.py gpurun_out/tl2/raw > gpurun_out/tl2/timeline.txt 2>&1
rm -rf gpurun_out/tl2/raw
cat gpurun_out/tl2/timeline.txt
