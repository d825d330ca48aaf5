# round 4, session 5: flat-row walk with its row operations on L2-resident lines (RAYEN_IOF_ABL=8, wrong results) against the shipped build
out=gpurun_out/r04k; mkdir -p $out
V=scripts/ubench/variants
for rep in 1 2; do
for lib in $V/librayen_base.so $V/librayen_mfma_pair_io_abl8.so; do
  for cfg in c5 c5r; do
    RAYEN_HIP_LIBRARY=$PWD/$lib timeout 300 python scripts/ubench/io_bench.py --config $cfg --batches 262144,1048576 2>&1 | tail -1 | sed "s/^/$cfg /"
  done
done
done 2>&1 | tee $out/timing.txt
