# round 4, session 5: issue priority of the two waves of a SIMD inside the tile walk (RAYEN_IO_PRIO) -- one box, two passes
out=gpurun_out/r04h; mkdir -p $out
V=scripts/ubench/variants
for rep in 1 2; do
for lib in $V/librayen_base.so $V/librayen_mfma_pair_io_prio1.so $V/librayen_mfma_pair_io_prio2.so $V/librayen_mfma_pair_io_prio3.so; do
  RAYEN_HIP_LIBRARY=$PWD/$lib timeout 300 python scripts/ubench/io_bench.py --config c3 --batches 262144,1048576 2>&1 | tail -1 | sed "s/^/c3 /"
  RAYEN_HIP_LIBRARY=$PWD/$lib timeout 300 python scripts/ubench/io_bench.py --config c5 --batches 262144,1048576 2>&1 | tail -1 | sed "s/^/c5 /"
done
done 2>&1 | tee $out/prio.txt
