#!/bin/bash
# A/B variant of the 32-lane kernels: tools/ab_rowk.sh <name> "<extra hipcc flags>" -> avx-ecm_amd/libgecm_<name>.so
# (only gecm_rowk.o is rebuilt; everything else is shared with the main build)
set -e
cd "$(dirname "$0")/../avx-ecm_amd"
name=$1; shift
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function "$@" -c csrc/gecm_rowk.hip -o build/ab_rowk_${name}.o
objs=$(ls build/gecm_kernels_*.o)
hipcc --offload-arch=gfx950 -shared -fPIC -o libgecm_${name}.so build/ab_rowk_${name}.o $objs build/gecm_dev.o build/gecm_api.o build/gecm_plan.o build/gecm_pair.o build/mpl.o build/calc_lite.o build/cunningham.o
