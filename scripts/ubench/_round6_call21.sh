out=gpurun_out/r06x; mkdir -p $out
V=$PWD/scripts/ubench/variants
timeout 300 python scripts/ubench/wl_check.py --batches 262144,1048576 2>&1 | grep -v amdgpu.ids > $out/wl_check.txt
echo "== nt1w12" >> $out/wl_check.txt
RAYEN_HIP_LIBRARY=$V/librayen_mfma_pair_wl_nt1w12.so timeout 300 python scripts/ubench/wl_check.py --batches 262144,1048576 2>&1 | grep -v amdgpu.ids >> $out/wl_check.txt
cat $out/wl_check.txt
RAYEN_HIP_LIBRARY=$V/librayen_mfma_pair_wl_stamps.so timeout 200 python scripts/ubench/wl_stamps.py 1048576 2>&1 | grep -v amdgpu.ids > $out/stamps.txt; cat $out/stamps.txt
