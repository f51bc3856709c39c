#!/bin/bash
# A/B of hwgrad_kernel's workgroup order (run ON the GPU box).  (The prefetch-distance variants this script also swept in
# round 3 -- SR3D_HWGRAD_PF -- are no longer built: within 1 %, profiles/r03a_ab_hwgrad_prefetch_distance_xcd_order.log.)
for dt in fp32 bf16; do
 for xcd in 0 1; do
  echo "== dtype $dt XCD $xcd"
  SR3D_HWGRAD_XCD=$xcd python tools/layer_bench.py --only up1 --wgrad-only --dtype $dt 2>&1 | grep -E "^up1|total wgrad"
 done
done
