out=$PWD/gpurun_out/r06m; mkdir -p $out
export TMPDIR=/tmp
for s in 3 1; do
cmd="python scripts/ubench/io_bench.py --schedule $s --batches 1048576 --reps 20"
rocprofv3 --kernel-trace --stats -d $out/stats$s -o stats -- $cmd > $out/stats$s.txt 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $out/pmc_a$s -o pmc -- $cmd > /dev/null 2> $out/pmc_a$s.err
rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_ANY -d $out/pmc_b$s -o pmc -- $cmd > /dev/null 2> $out/pmc_b$s.err
python scripts/ubench/pmc_dump.py $out/pmc_a$s pair > $out/pmc$s.txt 2>&1
python scripts/ubench/pmc_dump.py $out/pmc_b$s pair >> $out/pmc$s.txt 2>&1
grep -h "pair" $out/stats$s/*kernel_stats.csv $out/stats$s/*/*kernel_stats.csv 2>/dev/null | head -3 >> $out/pmc$s.txt
rm -rf $out/stats$s $out/pmc_a$s $out/pmc_b$s
done
cat $out/pmc3.txt $out/pmc1.txt
