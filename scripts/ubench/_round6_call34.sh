out=gpurun_out/r06zp; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_backward_dense_pairs.py -m gpu -q -x -s --timeout 600 -p no:cacheprovider > $out/pytest_bwdd.log 2>&1; tail -25 $out/pytest_bwdd.log
timeout 300 python scripts/ubench/bwd_bench.py c3 > $out/bwd_bench.txt 2>&1; RAYEN_BWD_DENSE_PAIRS=0 timeout 300 python scripts/ubench/bwd_bench.py c3 >> $out/bwd_bench.txt 2>&1; grep -v amdgpu $out/bwd_bench.txt
