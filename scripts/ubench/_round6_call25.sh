out=$PWD/gpurun_out/r06zc; mkdir -p $out
export TMPDIR=/tmp
cmd="python scripts/ubench/io_bench.py --schedule 3 --batches 1048576 --reps 20"
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $out/pmc_a -o pmc -- $cmd > /dev/null 2> $out/pmc_a.err
rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_ANY -d $out/pmc_b -o pmc -- $cmd > /dev/null 2> $out/pmc_b.err
rocprofv3 --pmc SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_IFETCH SQ_THREAD_CYCLES_VALU SQC_ICACHE_MISSES SQC_ICACHE_REQ -d $out/pmc_c -o pmc -- $cmd > /dev/null 2> $out/pmc_c.err
for d in a b c; do python scripts/ubench/pmc_dump.py $out/pmc_$d pair_wl >> $out/pmc.txt 2>&1; done
rm -rf $out/pmc_a $out/pmc_b $out/pmc_c
awk '{print $(NF-2), $(NF-1)}' $out/pmc.txt
tail -3 $out/pmc_c.err
