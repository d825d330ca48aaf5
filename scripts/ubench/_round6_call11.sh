out=gpurun_out/r06l; mkdir -p $out
for abl in 2 18 34 50 3; do
  RAYEN_HIP_LIBRARY=$PWD/scripts/ubench/variants/librayen_mfma_pair_wl_abl$abl.so timeout 200 python scripts/ubench/io_bench.py --schedule 3 --batches 262144,1048576 2>&1 | grep -v amdgpu.ids | sed "s/^/abl$abl: /" >> $out/abl.txt
done
cat $out/abl.txt
