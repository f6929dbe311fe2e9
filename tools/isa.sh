#!/bin/bash
# Device ISA of one kernel source: tools/isa.sh render.hip [out.s]   (authoring aid; hipcc cross-compiles without a GPU)
set -e
C=/root/repo/luisacomputegaussiansplatting_amd/csrc
OUT=${2:-/tmp/isa/${1%.hip}.s}
mkdir -p $(dirname $OUT)
cd $C && /opt/rocm/bin/hipcc -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize --offload-arch=gfx950 -I../../include -x hip --cuda-device-only -S kernels/$1 -o $OUT 2>&1 | grep -v "warning: argument unused" || true
echo $OUT
