#!/bin/bash
# A/B builds of sr3d_hconv_s2.hip with -D flags (built HERE by hipcc; the .so files travel with the snapshot).
# usage: bash tools/ab_s2_variants.sh build NAME:"-DFLAG=.. -DFLAG=.." ...
#        bash tools/ab_s2_variants.sh run DTYPE NAME...     (on the GPU box; per-layer table of the four stride-2 layers)
set -e
cd "$(dirname "$0")/.."
CS=3d-sr-micrometeorology_amd/csrc
mode=$1; shift
if [ "$mode" = build ]; then
  make -s -C $CS
  mkdir -p tools/abl
  for spec in "$@"; do
    n=${spec%%:*}; flags=${spec#*:}
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -fvisibility=hidden --offload-arch=gfx950 -Wno-unused-function -Wno-unused-variable $flags \
      -c $CS/sr3d_hconv_s2.hip -o tools/abl/s2_$n.o
    objs=$(ls $CS/*.o | grep -v sr3d_hconv_s2.o)
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fvisibility=hidden -o tools/abl/libsr3d_s2_$n.so $objs tools/abl/s2_$n.o
    rm -f tools/abl/s2_$n.o
  done
  exit 0
fi
dt=$1; shift
for n in "$@"; do
  echo "== $n ($dt)"
  for only in down1.0 down2.0 down3.0 down4.0; do
    SR3D_LIBRARY=$PWD/tools/abl/libsr3d_s2_$n.so python tools/layer_bench.py --only $only --dtype $dt --iters 5 2>/dev/null | grep "^down"
  done
done
