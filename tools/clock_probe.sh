#!/bin/bash
# samples the shader clock and power while a layer benchmark runs (run ON the GPU box): clock_probe.sh MODE LAYER
MODE=${1:-1}
LAYER=${2:-up1.up}
export SR3D_SPLIT_F16=$MODE
python3 tools/layer_bench.py --only $LAYER --iters 150 > gpurun_out/clock_probe_layers_$MODE.log 2>&1 &
PID=$!
while kill -0 $PID 2>/dev/null; do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|Power (W)" | sed 's/.*(\([0-9]*Mhz\)).*/\1/; s/.*Power (W): //' | tr '\n' ' '
  echo
  sleep 0.7
done
cat gpurun_out/clock_probe_layers_$MODE.log
