#!/bin/bash
# A/B: staging before / after the step's MFMAs in hwgrad_kernel (run ON the GPU box)
for dt in fp32 bf16; do
 for late in 0 1; do
   echo "== dtype $dt LATE $late"
   SR3D_HWGRAD_LATE=$late python tools/layer_bench.py --only up1 --wgrad-only --dtype $dt 2>&1 | grep -E "^up1|total wgrad"
 done
done
