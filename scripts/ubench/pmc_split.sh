#!/bin/bash
# counters of the split kernel for one variant library:  pmc_split.sh <variant> [bench args]
v=$1; shift
export TMPDIR=/tmp RAYEN_SPLIT_BF16=2 RAYEN_HIP_LIBRARY=$PWD/scripts/ubench/variants/librayen_$v.so
out=$PWD/gpurun_out/pmc_$v
rm -rf $out; mkdir -p $out
args="--steps 30 --warmup 5 --no-cpu-baseline $*"
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY -d $out/a -o pmc -- python bench.py $args > /dev/null 2> $out/a.err
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY SQ_WAIT_ANY -d $out/b -o pmc -- python bench.py $args > /dev/null 2> $out/b.err
python scripts/ubench/pmc_dump.py $out split
