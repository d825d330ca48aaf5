out=gpurun_out/r06zr; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_backward_dense_pairs.py tests/test_gpu_backward.py -m gpu -q -x --timeout 600 -p no:cacheprovider > $out/pytest_bwd.log 2>&1; tail -4 $out/pytest_bwd.log
echo "== new" > $out/bwd_bench.txt; timeout 300 python scripts/ubench/bwd_bench.py c3 2>&1 | grep -v amdgpu >> $out/bwd_bench.txt
echo "== old" >> $out/bwd_bench.txt; RAYEN_BWD_DENSE_PAIRS=0 timeout 300 python scripts/ubench/bwd_bench.py c3 2>&1 | grep -v amdgpu >> $out/bwd_bench.txt
cat $out/bwd_bench.txt
timeout 2300 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider --durations=60 > $out/pytest_gpu_durations.log 2>&1; tail -75 $out/pytest_gpu_durations.log
