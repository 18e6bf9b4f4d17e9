#!/bin/bash
# tools/build_variant_v1.sh NAME -DFLAG=VAL ...   -> nerf_few_shot_limitations_amd/libnerfhip_NAME.so
# A tuning variant of the V1 render kernels only (fused_v1.hip recompiled with the flags, every other object reused from the
# product build): load with NRF_LIB=<path> for same-box A/B runs.  Only for macros that do not change the packed stream layout.
set -e
NAME=$1; shift
PKG=nerf_few_shot_limitations_amd
mkdir -p $PKG/build/$NAME
cp $PKG/build/*.o $PKG/build/$NAME/
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -x hip -Wall -Wno-unused-function -Wno-unused-variable -fno-gpu-rdc -ffp-contract=off -Iinclude "$@" \
      -c $PKG/csrc/fused_v1.hip -o $PKG/build/$NAME/fused_v1.o
hipcc --offload-arch=gfx950 -shared -fPIC -o $PKG/libnerfhip_$NAME.so $PKG/build/$NAME/*.o
echo built $PKG/libnerfhip_$NAME.so
