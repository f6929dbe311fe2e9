# A/B of prebuilt libraries on the dense optimiser step alone: gpurun -- bash tools/gpu/ab_adam_libs.sh <tagA> <tagB> ...
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do for v in "$@"; do
cp gpurun_in/liblcgs_$v.so luisacomputegaussiansplatting_amd/liblcgs_hip.so
echo -n "$v "; timeout 200 python tools/gpu/adam_bench.py 2>/dev/null | tail -1
done; done
