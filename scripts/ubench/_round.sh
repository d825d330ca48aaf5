mkdir -p gpurun_out/r02k
timeout 900 python -m pytest tests/test_gpu_backward.py -q -k "bucketed" -p no:cacheprovider 2>&1 | tail -3
python scripts/ubench/bwd_split.py 2>&1 | grep -E "c3\"|2quad" | tee gpurun_out/r02k/bwd_split.log
