#!/bin/bash
# register / spill / scratch table of one translation unit:  tools/ra_unit.sh quad 9 1 [filter]   |   tools/ra_unit.sh hex 9
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/diag/asm
C=riemannian_motion_policies_amd/csrc
if [ "$1" = quad ]; then SRC=$C/rmp2_quad_tu.hip; DEFS="-DRMP2_TU_N=$2 -DRMP2_TU_SLOTS=$3"; FLT="${4:-quad_kernel}"; TAG=quad_n$2_s$3
else SRC=$C/rmp2_hex_tu.hip; DEFS="-DRMP2_TU_N=$2"; FLT="${3:-hex_kernel}"; TAG=hex_n$2; fi
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -fPIC $DEFS $EXTRA --cuda-device-only -S -Rpass-analysis=kernel-resource-usage -o tools/diag/asm/$TAG.s $SRC 2> tools/diag/asm/$TAG.remarks
python tools/resource_table.py tools/diag/asm/$TAG.remarks "$FLT"
