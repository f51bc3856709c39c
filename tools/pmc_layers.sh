#!/bin/bash
# PMC counters (two passes) over tools/layer_bench.py --only $1 (run ON the GPU box): tools/pmc_layers.sh up1 TAG
set -u
SEL=${1:-up1}; TAG=${2:-x}
OUT=gpurun_out/pmc_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
L="python3 tools/layer_bench.py --iters 1"
rocprofv3 --pmc SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM --output-format csv -d $OUT/p1 -o a -- $L --only $SEL > $OUT/p1.log 2>&1 || exit 1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_COEXEC_CYCLES --output-format csv -d $OUT/p2 -o b -- $L --only $SEL > $OUT/p2.log 2>&1 || exit 1
python3 tools/pmc_mix.py $OUT/pmc_instruction_mix_$SEL.json $OUT/p1 $OUT/p2
rm -rf $OUT/p1 $OUT/p2
