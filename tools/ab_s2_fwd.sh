#!/bin/bash
# A/B of the stride-2 forward: pair form (both x parities per load) vs class form (SR3D_HCONV_S2_CLASS_FWD=1), per layer
set -e
for dt in fp32 bf16; do
  for only in down1.0 down2.0 down3.0 down4.0; do
    echo "== $dt $only pair form"; python tools/layer_bench.py --only $only --dtype $dt --iters 5 | tail -n +2 | head -1
    echo "== $dt $only class form"; SR3D_HCONV_S2_CLASS_FWD=1 python tools/layer_bench.py --only $only --dtype $dt --iters 5 | tail -n +2 | head -1
  done
done
