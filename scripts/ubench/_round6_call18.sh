out=gpurun_out/r06u; mkdir -p $out
V=$PWD/scripts/ubench/variants
for v in nt1w8 nt1w12 nt1w16; do
  echo "== $v" >> $out/wl_check.txt
  RAYEN_HIP_LIBRARY=$V/librayen_mfma_pair_wl_$v.so timeout 300 python scripts/ubench/wl_check.py --batches 262144,1048576 2>&1 | grep -v amdgpu.ids >> $out/wl_check.txt
done
cat $out/wl_check.txt
