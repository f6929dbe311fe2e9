# One forward-only frame of the spatially ordered scene as a kernel timeline: gpurun -- bash tools/gpu/timeline_forward_spatial.sh
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
export LCGS_BENCH_SPATIAL_FIRST=1
mkdir -p gpurun_out/tls
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tls/raw -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-train-step --no-stage-path --no-backward --no-batch --no-spatial > gpurun_out/tls/bench.json 2> gpurun_out/tls/err.log
python3 profiles/timeline.py gpurun_out/tls/raw > gpurun_out/tls/timeline.txt 2>&1
rm -rf gpurun_out/tls/raw
tail -40 gpurun_out/tls/timeline.txt
