#!/bin/bash
set -u
OUT=gpurun_out/prof_hconv
mkdir -p $OUT
export TMPDIR=/tmp SR3D_SPLIT_F16=1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python3 tools/layer_bench.py --only up1 --iters 3 > $OUT/trace.log 2>&1 || exit 1
find $OUT/trace -name "*kernel_stats.csv" -exec cp {} $OUT/hconv_up1_kernel_stats.csv \;
rm -rf $OUT/trace
cat $OUT/trace.log | tail -8
cut -c1-200 $OUT/hconv_up1_kernel_stats.csv | head -30
