#!/bin/bash
# tools/build_variant.sh NAME -DFLAG=VAL ...   -> nerf_few_shot_limitations_amd/libnerfhip_NAME.so
# A tuning variant of the fused render / forward kernels of every network family (fused_v1 ... fused_v3w recompiled with the
# flags, in parallel; every other object reused from the product build): load with NRF_LIB=<path> for same-box A/B runs.
# Only for macros that do not change the packed stream layout.
set -e
NAME=$1; shift
PKG=nerf_few_shot_limitations_amd
mkdir -p $PKG/build/$NAME
cp $PKG/build/*.o $PKG/build/$NAME/
for f in fused_v1 fused_v2 fused_v3 fused_v3w; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -x hip -Wall -Wno-unused-function -Wno-unused-variable -fno-gpu-rdc -ffp-contract=off -Iinclude "$@" \
        -c $PKG/csrc/$f.hip -o $PKG/build/$NAME/$f.o &
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o $PKG/libnerfhip_$NAME.so $PKG/build/$NAME/*.o
echo built $PKG/libnerfhip_$NAME.so
