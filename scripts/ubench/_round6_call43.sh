out=gpurun_out/r06zz; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_backward_dense_pairs.py tests/test_gpu_backward_pairs.py -m gpu -q -x --timeout 600 -p no:cacheprovider > $out/pytest_bwd.log 2>&1; tail -4 $out/pytest_bwd.log
for i in 1 2; do timeout 300 python scripts/ubench/bwd_bench.py c3 2>&1 | grep -v amdgpu | tail -4; done
