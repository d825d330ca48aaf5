set -x
out=gpurun_out/r06c; mkdir -p $out
RAYEN_HIP_LIBRARY=$PWD/scripts/ubench/variants/librayen_mfma_pair_io_iostamps.so timeout 300 python scripts/ubench/io_stamps.py > $out/io_stamps.txt 2>&1; cat $out/io_stamps.txt
timeout 600 python -m pytest tests/test_gpu_feasibility.py -m gpu -q --timeout 600 -p no:cacheprovider -s > $out/pytest_a.log 2>&1; tail -3 $out/pytest_a.log; grep "rows > 0\|clipped" $out/pytest_a.log
