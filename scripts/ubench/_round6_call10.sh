out=gpurun_out/r06k; mkdir -p $out
timeout 200 python scripts/ubench/io_bench.py --schedule 3 --batches 262144,1048576 2>&1 | grep -v amdgpu.ids | sed "s/^/w12: /" >> $out/abl.txt
for v in w16 w8 w16ns; do
  RAYEN_HIP_LIBRARY=$PWD/scripts/ubench/variants/librayen_mfma_pair_wl_$v.so timeout 200 python scripts/ubench/io_bench.py --schedule 3 --batches 262144,1048576 2>&1 | grep -v amdgpu.ids | sed "s/^/$v: /" >> $out/abl.txt
done
cat $out/abl.txt
