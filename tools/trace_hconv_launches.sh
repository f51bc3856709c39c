#!/bin/bash
# per-launch durations of hconv_kernel over one training step (rocprofv3 --kernel-trace), run ON the GPU box:
#   bash tools/trace_hconv_launches.sh TAG   -> gpurun_out/TAG_hconv_launches.txt   (env is inherited: A/B by setting switches)
set -u
TAG=$1
OUT=gpurun_out/trace_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/t -o k -- python3 bench.py --no-cpu-baseline --no-secondary --steps 1 --warmup 1 > $OUT/log.txt 2>&1 || exit 1
python3 - <<PY
import csv,glob
f=glob.glob("$OUT/t/**/*kernel_trace.csv", recursive=True)[0]
rows=[r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
h=[(r["Kernel_Name"], (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6, r.get("Grid_Size_X", r.get("Grid_Size","")), r.get("Workgroup_Size_X","")) for r in rows if "hconv_kernel" in r["Kernel_Name"]]
n=len(h)//4          # the run has 4 steps; take the last one
last=h[-n:]
with open("gpurun_out/${TAG}_hconv_launches.txt","w") as o:
    for i,(k,ms,g,w) in enumerate(last): o.write(f"{i:3d} {ms:8.3f} ms grid {g} {k[28:60]}\n")
    o.write(f"total {sum(x[1] for x in last):.2f} ms over {n} launches\n")
PY
rm -rf $OUT/t
