out=gpurun_out/r06zm; mkdir -p $out
V=$PWD/scripts/ubench/variants
for v in p16 p12; do
echo "== $v" >> $out/wl_check.txt
RAYEN_HIP_LIBRARY=$V/librayen_mfma_pair_wl_$v.so timeout 300 python scripts/ubench/wl_check.py --batches 262144,200001,1048576 2>&1 | grep -v amdgpu.ids >> $out/wl_check.txt
done
echo "== base" >> $out/wl_check.txt
timeout 300 python scripts/ubench/wl_check.py --batches 262144,1048576 2>&1 | grep -v amdgpu.ids >> $out/wl_check.txt
cat $out/wl_check.txt
