#!/bin/bash
# PMC counters of the split-f16 conv kernel on the level-0/1 decoder layers (run ON the GPU box)
set -u
OUT=gpurun_out/prof_hconv
mkdir -p $OUT
export TMPDIR=/tmp SR3D_SPLIT_F16=1
L="python3 tools/layer_bench.py --iters 1"
rocprofv3 --pmc SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM --output-format csv -d $OUT/pmc_mix1 -o a -- $L --only up1 > $OUT/pmc_mix1.log 2>&1 || exit 1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_COEXEC_CYCLES --output-format csv -d $OUT/pmc_mix2 -o b -- $L --only up1 > $OUT/pmc_mix2.log 2>&1 || exit 1
python3 tools/pmc_mix.py $OUT/pmc_instruction_mix_hconv_up1.json $OUT/pmc_mix1 $OUT/pmc_mix2
rm -rf $OUT/pmc_mix1 $OUT/pmc_mix2
