#!/bin/bash
# Knock-out timing of k_render_backward: builds liblcgs_ko{1,2,3}.so (-DLCGS_BWD_KO: 1 no reduction, 2 no evaluation, 3 no flush;
# wrong results, timing only) into gpurun_in/ next to a copy of the current library, for
#   gpurun -- bash tools/gpu/ab_bwd.sh cur ko1 ko2 ko3
set -e
R=$(cd $(dirname $0)/../.. && pwd); C=$R/luisacomputegaussiansplatting_amd/csrc
make -C $C -j8 >/dev/null
mkdir -p $R/gpurun_in && cp $R/luisacomputegaussiansplatting_amd/liblcgs_hip.so $R/gpurun_in/liblcgs_cur.so
FLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -fvisibility=hidden -Wall -Wno-unused-result --offload-arch=gfx950 -I$R/include"
OTHERS=$(ls $C/build/kernels/*.o $C/build/*.o $C/build/host/*.o | grep -v backward.hip.o)
for k in 1 2 3; do
  /opt/rocm/bin/hipcc $FLAGS -DLCGS_BWD_KO=$k -x hip -c $C/kernels/backward.hip -o /tmp/lcgs_ko$k.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/gpurun_in/liblcgs_ko$k.so /tmp/lcgs_ko$k.o $OTHERS -lpthread -ldl
done
ls -la $R/gpurun_in/
