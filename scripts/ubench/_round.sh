# round 4, last session: config 3's walk with its A re-loads kept but the counted waits removed / relaxed (wrong results) -- is it the waits or the loads?
out=gpurun_out/r04w; mkdir -p $out
V=scripts/ubench/variants
for rep in 1 2; do
for lib in rayen_amd/csrc/librayen_hip.so $V/librayen_mfma_pair_io_nowait.so $V/librayen_mfma_pair_io_latewait.so; do
  RAYEN_HIP_LIBRARY=$PWD/$lib timeout 300 python scripts/ubench/io_bench.py --config c3 --batches 262144,1048576 2>&1 | tail -1 | sed "s/^/c3 /"
done
done 2>&1 | tee $out/waits.txt
