#!/bin/bash
# timing-only ablation builds of hconv_kernel in bf16 storage (results are WRONG by construction)
echo "== full"; python tools/layer_bench.py --only up1 --dtype bf16 2>&1 | grep -E "^up1|total"
for n in 1 2 3 4; do
  echo "== HCONV_ABL=$n (1: no halo refill, 2: no weight DMA, 3: no barriers, 4: no LDS weight-fragment reads)"
  SR3D_LIBRARY=$PWD/tools/abl/libsr3d_abl$n.so python tools/layer_bench.py --only up1 --dtype bf16 2>&1 | grep -E "^up1|total"
done
