# round 4, session 5: the flat-row kernel's runs of linear tiles -- bit-equality tests, then timings of three builds on one box
out=gpurun_out/r04e; mkdir -p $out
timeout 1500 python -m pytest tests/test_gpu_pair_io.py -m gpu -q -k "flat" --timeout 900 -p no:cacheprovider -x > $out/pytest_flat.log 2>&1
tail -4 $out/pytest_flat.log
V=scripts/ubench/variants
for rep in 1 2; do
for lib in $V/librayen_base.so $V/librayen_mfma_pair_io_noruns.so rayen_amd/csrc/librayen_hip.so; do
  for cfg in c5 c5r; do
    RAYEN_HIP_LIBRARY=$PWD/$lib timeout 300 python scripts/ubench/io_bench.py --config $cfg --batches 262144,524288 2>&1 | tail -1 | sed "s/^/$cfg /"
    RAYEN_HIP_LIBRARY=$PWD/$lib timeout 300 python scripts/ubench/io_bench.py --config $cfg --batches 262144 --track 2>&1 | tail -1 | sed "s/^/$cfg track /"
  done
done
done 2>&1 | tee $out/timing.txt
