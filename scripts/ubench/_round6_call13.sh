out=gpurun_out/r06o; mkdir -p $out
timeout 300 python scripts/ubench/wl_check.py --batches 262144,200001,1048576 > $out/wl_check.txt 2>&1; cat $out/wl_check.txt
for v in nt1w12 nt2ns; do
  RAYEN_HIP_LIBRARY=$PWD/scripts/ubench/variants/librayen_mfma_pair_wl_$v.so timeout 200 python scripts/ubench/io_bench.py --schedule 3 --batches 262144,1048576 2>&1 | grep -v amdgpu.ids | sed "s/^/$v: /" >> $out/abl.txt
done
cat $out/abl.txt
