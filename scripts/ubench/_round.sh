mkdir -p gpurun_out/r02l
timeout 900 python -m pytest tests/test_gpu_backward.py -q -k "bucketed" -p no:cacheprovider 2>&1 | tail -8
BWD_FP64=1 python scripts/ubench/bwd_bench.py c3 2>&1 | tail -2
python scripts/ubench/bwd_bench.py c3 c2 c5 2>&1 | tail -3
