#!/bin/bash
# tools/build_variant.sh NAME -DFLAG=VAL ...   -> nerf_few_shot_limitations_amd/libnerfhip_NAME.so
# A tuning variant of the fused render / forward kernels (every family, both mode halves recompiled with the extra -D flags; every
# other object reused from the product build): load with NRF_LIB=<path> for same-box A/B runs (tools/ab_variants.sh).
# Only for macros that do not change the packed stream layout.
set -e
NAME=$1; shift
python -m nerf_few_shot_limitations_amd.build --variant "$NAME" --fused-only "$@"
