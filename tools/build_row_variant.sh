#!/bin/bash
# Build avx-ecm_amd/libgecm_<name>.so: the shipped objects with the 32-lane stage-1 object (csrc/gecm_rowk.hip)
# recompiled with extra flags, for A/B runs with tools/ab_row_libs.py (GECM_LIB picks the library).
# usage: tools/build_row_variant.sh <name> "<extra hipcc flags>"      e.g.  prio8 "-DGECM_ROW_WG_WAVES=8 -DGECM_ROW_PRIO=3"
set -e
cd "$(dirname "$0")/../avx-ecm_amd"
name=$1; shift
make -s -j8 libgecm.so >/dev/null
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -DGECM_MANIFEST='"variant-'$name'"' $* \
      -c csrc/gecm_rowk.hip -o build/gecm_rowk_$name.o
objs=$(ls build/*.o | grep -v 'gecm_rowk' | grep -v avx_ecm_main)
hipcc --offload-arch=gfx950 -shared -fPIC -o libgecm_$name.so $objs build/gecm_rowk_$name.o
echo "built avx-ecm_amd/libgecm_$name.so ($*)"
