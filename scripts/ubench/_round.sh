# round 4, last binary: the random-set fuzz at 1000 seeds (forward parity + backward), as on round 2's final binary
out=gpurun_out/r04x; mkdir -p $out
RAYEN_FUZZ_SEEDS=1000 timeout 2700 python -m pytest tests/test_gpu_parity.py tests/test_gpu_backward.py -m gpu -q -k "random" --timeout 900 -p no:cacheprovider > $out/fuzz1000.log 2>&1
tail -6 $out/fuzz1000.log | cut -c1-300
