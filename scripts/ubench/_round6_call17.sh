out=gpurun_out/r06s; mkdir -p $out
V=$PWD/scripts/ubench/variants
timeout 300 python scripts/ubench/wl_check.py --batches 262144,200001,1048576 > $out/wl_check.txt 2>&1; cat $out/wl_check.txt
RAYEN_HIP_LIBRARY=$V/librayen_mfma_pair_wl_clock.so timeout 200 python scripts/ubench/wl_clock.py --schedule 3 2>&1 | grep -v amdgpu.ids | sed "s/^/wl: /" >> $out/clock.txt
cat $out/clock.txt
