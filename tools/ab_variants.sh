#!/bin/bash
# same-box A/B of library variants (tools/build_variant.sh): tools/ab_variants.sh base pf2 pf5 ...
for v in "$@"; do
  lib=nerf_few_shot_limitations_amd/libnerfhip_$v.so
  [ "$v" = base ] && lib=nerf_few_shot_limitations_amd/libnerfhip.so
  for mode in bf16 f16x3; do
    NRF_LIB=$PWD/$lib python bench.py --mode $mode --steps 8 --warmup 2 --no-train --no-extras --cpu-rows 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', '$mode', d['roofline']['kernel_ms'], d['roofline']['frac'])"
  done
done
