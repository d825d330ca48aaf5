out=gpurun_out/r06zs; mkdir -p $out
timeout 1500 python -m pytest tests/test_gpu_backward_dense_pairs.py tests/test_gpu_backward_pairs.py "tests/test_gpu_backward.py::test_bucketed_backward_equals_the_plain_walk" "tests/test_gpu_parity.py::test_batches_beyond_2_31_elements" -m gpu -q --timeout 900 -p no:cacheprovider > $out/pytest_bwd.log 2>&1; tail -6 $out/pytest_bwd.log
echo "== new" > $out/bwd_bench.txt; timeout 300 python scripts/ubench/bwd_bench.py c3 2>&1 | grep -v amdgpu >> $out/bwd_bench.txt
echo "== old" >> $out/bwd_bench.txt; RAYEN_BWD_DENSE_PAIRS=0 timeout 300 python scripts/ubench/bwd_bench.py c3 2>&1 | grep -v amdgpu >> $out/bwd_bench.txt
echo "== new" >> $out/bwd_bench.txt; timeout 300 python scripts/ubench/bwd_bench.py c3 2>&1 | grep -v amdgpu >> $out/bwd_bench.txt
cat $out/bwd_bench.txt
