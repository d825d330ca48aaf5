out=gpurun_out/r06p; mkdir -p $out
V=$PWD/scripts/ubench/variants
RAYEN_HIP_LIBRARY=$V/librayen_mfma_pair_wl_clock.so timeout 200 python scripts/ubench/wl_clock.py --schedule 3 2>&1 | grep -v amdgpu.ids | sed "s/^/wl: /" >> $out/clock.txt
RAYEN_HIP_LIBRARY=$V/librayen_mfma_pair_wl_clockns.so timeout 200 python scripts/ubench/wl_clock.py --schedule 3 2>&1 | grep -v amdgpu.ids | sed "s/^/wl no stores: /" >> $out/clock.txt
RAYEN_HIP_LIBRARY=$V/librayen_mfma_pair_wl_clocknm.so timeout 200 python scripts/ubench/wl_clock.py --schedule 3 2>&1 | grep -v amdgpu.ids | sed "s/^/wl no stores no MFMA: /" >> $out/clock.txt
RAYEN_HIP_LIBRARY=$V/librayen_mfma_pair_io_clock.so timeout 200 python scripts/ubench/wl_clock.py --schedule 1 2>&1 | grep -v amdgpu.ids | sed "s/^/io: /" >> $out/clock.txt
cat $out/clock.txt
