out=gpurun_out/r06j; mkdir -p $out
for abl in 8 12 2 10; do
  RAYEN_HIP_LIBRARY=$PWD/scripts/ubench/variants/librayen_mfma_pair_wl_abl$abl.so timeout 200 python scripts/ubench/io_bench.py --schedule 3 --batches 262144,1048576 2>&1 | grep -v amdgpu.ids | sed "s/^/abl$abl: /" >> $out/abl.txt
done
cat $out/abl.txt
