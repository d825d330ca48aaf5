#!/bin/bash
# Profile `bench.py` on the GPU box: kernel-trace stats first, then PMC passes (separately, as
# gpurun requires).  Outputs land in gpurun_out/prof_<tag>/; copy the summaries into profiles/.
#   scripts/profile_bench.sh <tag> [bench args...]
set -u
tag=${1:-r01}; shift || true
out=$PWD/gpurun_out/prof_$tag
mkdir -p "$out"
export TMPDIR=/tmp
args="--steps 200 --warmup 20 --no-cpu-baseline $*"
rocprofv3 --kernel-trace --stats -d "$out/stats" -o stats -- python bench.py $args > "$out/bench_stats.json" 2> "$out/stats.err"
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d "$out/pmc_sq" -o pmc -- python bench.py $args > /dev/null 2> "$out/pmc_sq.err"
rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE -d "$out/pmc_fetch" -o pmc -- python bench.py $args > /dev/null 2> "$out/pmc_fetch.err"
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d "$out/pmc_write" -o pmc -- python bench.py $args > /dev/null 2> "$out/pmc_write.err"
find "$out" -name "*.csv" | head -40
