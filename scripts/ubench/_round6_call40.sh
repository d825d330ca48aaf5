out=gpurun_out/r06zv; mkdir -p $out
timeout 1500 python -m pytest tests/test_gpu_backward_dense_pairs.py -m gpu -q --timeout 900 -p no:cacheprovider > $out/pytest_bwd.log 2>&1; tail -3 $out/pytest_bwd.log
for i in 1 2; do
echo "== new" >> $out/bwd_bench.txt; timeout 300 python scripts/ubench/bwd_bench.py c3 2>&1 | grep -v amdgpu >> $out/bwd_bench.txt
echo "== old" >> $out/bwd_bench.txt; RAYEN_BWD_DENSE_PAIRS=0 timeout 300 python scripts/ubench/bwd_bench.py c3 2>&1 | grep -v amdgpu >> $out/bwd_bench.txt
done
cat $out/bwd_bench.txt
RAYEN_HIP_LIBRARY=$PWD/scripts/ubench/variants/librayen_mfma_bwdd_stamps.so timeout 200 python scripts/ubench/bwdd_stamps.py 2>&1 | grep -v amdgpu | head -2
