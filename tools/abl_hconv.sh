#!/bin/bash
# Timing-only ablation builds of hconv_kernel (results are WRONG by construction).  HCONV_ABL is a bit mask:
#   1 no halo refill, 2 no weight DMA, 4 no barriers, 8 no weight-fragment LDS reads, 16 no halo-fragment LDS reads,
#   32 one MFMA of the three products
# usage: bash tools/abl_hconv.sh build MASK...      (here: hipcc cross-compiles; the .so files travel with the snapshot)
#        bash tools/abl_hconv.sh run [fp32|bf16] MASK...   (on the GPU box)
set -e
cd "$(dirname "$0")/.."
CS=3d-sr-micrometeorology_amd/csrc
mode=$1; shift
if [ "$mode" = build ]; then
  make -s -C $CS
  mkdir -p tools/abl
  for n in "$@"; do
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -fvisibility=hidden --offload-arch=gfx950 -Wno-unused-function -Wno-unused-variable -DHCONV_ABL=$n \
      -c $CS/sr3d_hconv.hip -o tools/abl/hconv_abl$n.o
    objs=$(ls $CS/*.o | grep -v sr3d_hconv.o)
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/abl/libsr3d_habl$n.so $objs tools/abl/hconv_abl$n.o
    rm -f tools/abl/hconv_abl$n.o
  done
  exit 0
fi
dt=$1; shift
echo "== full"; python tools/layer_bench.py --only up1 --dtype $dt 2>&1 | grep -E "^up1|total (fwd|dgrad)"
for n in "$@"; do
  echo "== HCONV_ABL=$n"
  SR3D_LIBRARY=$PWD/tools/abl/libsr3d_habl$n.so python tools/layer_bench.py --only up1 --dtype $dt 2>&1 | grep -E "^up1|total (fwd|dgrad)"
done
