out=gpurun_out/r06zt; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_pair_io.py -m gpu -q -x --timeout 600 -p no:cacheprovider > $out/pytest_io.log 2>&1; tail -4 $out/pytest_io.log
for i in 1 2; do timeout 300 python scripts/ubench/io_bench.py --config c5 --batches 262144,1048576 2>&1 | grep -v amdgpu | tail -1 >> $out/iof_nodummy.txt; done
timeout 300 python scripts/ubench/io_bench.py --config c5r --batches 262144 2>&1 | grep -v amdgpu | tail -1 >> $out/iof_nodummy.txt
cat $out/iof_nodummy.txt
