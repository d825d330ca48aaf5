#!/bin/bash
# Profile `bench.py` on the GPU box: kernel-trace stats first, then PMC passes (separately, as
# gpurun requires).  Outputs land in gpurun_out/prof_<tag>/; scripts/summarize_profile.py condenses
# them into profiles/<round>_<config>_..._rocprofv3.json (which bench.py reads back for `traffic`).
#   scripts/profile_bench.sh <tag> [bench args...]      e.g.  scripts/profile_bench.sh r02_c4 --config c4
set -u
tag=${1:-r02}; shift || true
out=$PWD/gpurun_out/prof_$tag
mkdir -p "$out"
export TMPDIR=/tmp
args="--steps 200 --warmup 20 --no-cpu-baseline --no-families --graph off $*"
rocprofv3 --kernel-trace --stats -d "$out/stats" -o stats -- python bench.py $args > "$out/bench_stats.json" 2> "$out/stats.err"
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d "$out/pmc_sq" -o pmc -- python bench.py $args > /dev/null 2> "$out/pmc_sq.err"
rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU -d "$out/pmc_sq2" -o pmc -- python bench.py $args > /dev/null 2> "$out/pmc_sq2.err"
rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE -d "$out/pmc_fetch" -o pmc -- python bench.py $args > /dev/null 2> "$out/pmc_fetch.err"
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d "$out/pmc_write" -o pmc -- python bench.py $args > /dev/null 2> "$out/pmc_write.err"
ls "$out"
