out=gpurun_out/r06zt; mkdir -p $out
timeout 1500 python -m pytest tests/test_gpu_backward_dense_pairs.py -m gpu -q -s --timeout 900 -p no:cacheprovider > $out/pytest_bwd.log 2>&1; grep "worst gradient" $out/pytest_bwd.log; tail -4 $out/pytest_bwd.log
for i in 1 2; do
echo "== new" >> $out/bwd_bench.txt; timeout 300 python scripts/ubench/bwd_bench.py c3 2>&1 | grep -v amdgpu >> $out/bwd_bench.txt
echo "== old" >> $out/bwd_bench.txt; RAYEN_BWD_DENSE_PAIRS=0 timeout 300 python scripts/ubench/bwd_bench.py c3 2>&1 | grep -v amdgpu >> $out/bwd_bench.txt
done
cat $out/bwd_bench.txt
