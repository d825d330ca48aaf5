# round 4, session 5: row operations of config 3's walk in batches (RAYEN_IO_BATCH) -- bit-equality with the plain pair kernel per batch size, then timing on one box
out=gpurun_out/r04m; mkdir -p $out
V=scripts/ubench/variants
for lib in $V/librayen_mfma_pair_io_batch2.so rayen_amd/csrc/librayen_hip.so; do
  RAYEN_HIP_LIBRARY=$PWD/$lib timeout 900 python -m pytest tests/test_gpu_pair_io.py -m gpu -q -k "not flat and not lds" --timeout 600 -p no:cacheprovider -x 2>&1 | tail -2 | sed "s|^|$lib |"
done 2>&1 | tee $out/pytest.txt
for rep in 1 2; do
for lib in $V/librayen_base.so $V/librayen_mfma_pair_io_batch1.so $V/librayen_mfma_pair_io_batch2.so rayen_amd/csrc/librayen_hip.so; do
  RAYEN_HIP_LIBRARY=$PWD/$lib timeout 300 python scripts/ubench/io_bench.py --config c3 --batches 262144,1048576 2>&1 | tail -1 | sed "s/^/c3 /"
done
done 2>&1 | tee $out/timing.txt
