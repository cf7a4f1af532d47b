#!/bin/bash
# Build an A/B variant of libgecm for kernel experiments: only the NL=15 kernel object is rebuilt
# with the given extra flags; everything else is shared with the main build.
#   tools/ab_build.sh <name> "<extra hipcc flags>"   ->  avx-ecm_amd/libgecm_<name>.so
set -e
cd "$(dirname "$0")/../avx-ecm_amd"
name=$1; shift
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -DGECM_NL=15 $@ -c csrc/gecm_kernels.hip -o build/ab_${name}_15.o
objs=$(ls build/gecm_kernels_*.o | grep -v "_15_p")
hipcc --offload-arch=gfx950 -shared -fPIC -o libgecm_${name}.so build/ab_${name}_15.o $objs build/gecm_dev.o build/gecm_api.o build/gecm_plan.o build/gecm_pair.o build/mpl.o build/calc_lite.o build/cunningham.o
ls -la libgecm_${name}.so
