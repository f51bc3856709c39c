#!/bin/bash
# one rocprofv3 --kernel-trace --stats pass over bench.py (run ON the GPU box): tools/stats_once.sh TAG [bench args]
set -u
TAG=$1; shift
OUT=gpurun_out/stats_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -o s -- python3 bench.py --no-cpu-baseline --no-secondary --steps 3 --warmup 1 "$@" > $OUT/log.txt 2>&1 || exit 1
find $OUT/t -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
rm -rf $OUT/t
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$OUT/kernel_stats.csv")))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print('total kernel ms per step', tot/4/1e6)
for r in rows[:26]:
    print(f"{float(r['TotalDurationNs'])/4/1e6:8.2f} ms/step {int(r['Calls'])/4:6.1f} calls  {r['Name'][:100]}")
PY
