#!/bin/bash
# Profiles of one round, run ON the GPU box from the repository root:  bash tools/collect_profiles.sh r02
# (rocprofv3 is given the program itself -- python3 ... -- never a wrapper; PMC passes carry no trace options)
set -u
TAG=${1:-r02}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
B="python3 bench.py --no-cpu-baseline --no-secondary"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_config1 -o c1 -- $B --steps 3 --warmup 1 > $OUT/stats_config1.log 2>&1 || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_config2 -o c2 -- $B --steps 2 --warmup 1 --batch 4 --loss mixed > $OUT/stats_config2.log 2>&1 || exit 1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o f -- $B --steps 1 --warmup 1 > $OUT/pmc_fetch.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o w -- $B --steps 1 --warmup 1 > $OUT/pmc_write.log 2>&1 || exit 1
L="python3 tools/layer_bench.py --iters 1"
rocprofv3 --pmc SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM --output-format csv -d $OUT/pmc_mix1 -o a -- $L --only 1 > $OUT/pmc_mix1.log 2>&1 || exit 1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_COEXEC_CYCLES --output-format csv -d $OUT/pmc_mix2 -o b -- $L --only 1 > $OUT/pmc_mix2.log 2>&1 || exit 1
python3 tools/pmc_traffic.py $OUT/pmc_fetch $OUT/pmc_write $OUT/${TAG}_pmc_hbm_traffic.json > /dev/null
python3 tools/pmc_mix.py $OUT/${TAG}_pmc_instruction_mix_level01_layers.json $OUT/pmc_mix1 $OUT/pmc_mix2
cp $OUT/stats_config1/c1_kernel_stats.csv $OUT/${TAG}_rocprofv3_kernel_stats_config1_steps3_warmup1.csv 2>/dev/null || find $OUT/stats_config1 -name "*kernel_stats.csv" -exec cp {} $OUT/${TAG}_rocprofv3_kernel_stats_config1_steps3_warmup1.csv \;
cp $OUT/stats_config2/c2_kernel_stats.csv $OUT/${TAG}_rocprofv3_kernel_stats_config2_batch4_mixed_steps2_warmup1.csv 2>/dev/null || find $OUT/stats_config2 -name "*kernel_stats.csv" -exec cp {} $OUT/${TAG}_rocprofv3_kernel_stats_config2_batch4_mixed_steps2_warmup1.csv \;
# the raw per-dispatch CSVs are large: keep only the summaries
rm -rf $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_mix1 $OUT/pmc_mix2 $OUT/stats_config1 $OUT/stats_config2
ls -la $OUT
