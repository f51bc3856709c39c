#!/bin/bash
# A/B of hwgrad_kernel's prefetch distance and workgroup order (run ON the GPU box)
for dt in fp32 bf16; do
 for pf in 0 1 2; do
  for xcd in 0 1; do
   echo "== dtype $dt PF $pf XCD $xcd"
   SR3D_HWGRAD_PF=$pf SR3D_HWGRAD_XCD=$xcd python tools/layer_bench.py --only up1 --wgrad-only --dtype $dt 2>&1 | grep -E "^up1|total wgrad"
  done
 done
done
