mkdir -p gpurun_out/r02n
RAYEN_FUZZ_SEEDS=1000 timeout 5000 python -m pytest tests/test_gpu_parity.py tests/test_gpu_backward.py -m gpu -q -k "random" --timeout 900 -p no:cacheprovider > gpurun_out/r02n/fuzz1000.log 2>&1
tail -12 gpurun_out/r02n/fuzz1000.log | cut -c1-300
