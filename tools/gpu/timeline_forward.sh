# One forward-only frame as a kernel timeline (rocprofv3 --kernel-trace): gpurun -- bash tools/gpu/timeline_forward.sh
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/tl
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl/raw -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-train-step --no-stage-path --no-backward --no-batch > gpurun_out/tl/bench.json 2> gpurun_out/tl/err.log

python3 profiles/timeline.py gpurun_out/tl/raw > gpurun_out/tl/timeline.txt 2>&1
rm -rf gpurun_out/tl/raw
tail -60 gpurun_out/tl/timeline.txt
