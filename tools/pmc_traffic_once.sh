#!/bin/bash
# HBM traffic per kernel (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes), run ON the GPU box:
#   bash tools/pmc_traffic_once.sh TAG [bench args]      -> gpurun_out/TAG_pmc_hbm_traffic.json
set -u
TAG=$1; shift
OUT=gpurun_out/pmc_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
B="python3 bench.py --no-cpu-baseline --no-secondary --steps 1 --warmup 1"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/f -o f -- $B "$@" > $OUT/fetch.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/w -o w -- $B "$@" > $OUT/write.log 2>&1 || exit 1
python3 tools/pmc_traffic.py $OUT/f $OUT/w gpurun_out/${TAG}_pmc_hbm_traffic.json > /dev/null
rm -rf $OUT/f $OUT/w
