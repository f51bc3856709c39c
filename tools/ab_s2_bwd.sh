#!/bin/bash
# A/B of the stride-2 input gradient: both x classes per workgroup vs class form (SR3D_HCONV_S2_CLASS_BWD=1), per layer
set -e
for dt in fp32 bf16; do
  for only in down1.0 down2.0 down3.0 down4.0; do
    echo "== $dt $only pair form";  python tools/layer_bench.py --only $only --dtype $dt --iters 5 2>/dev/null | grep "^down"
    echo "== $dt $only class form"; SR3D_HCONV_S2_CLASS_BWD=1 python tools/layer_bench.py --only $only --dtype $dt --iters 5 2>/dev/null | grep "^down"
  done
done
