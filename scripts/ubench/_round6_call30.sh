out=gpurun_out/r06zk; mkdir -p $out
V=$PWD/scripts/ubench/variants
for v in cbase cp12 cp12ns; do
echo "== $v" >> $out/clock.txt
RAYEN_HIP_LIBRARY=$V/librayen_mfma_pair_wl_$v.so timeout 300 python scripts/ubench/wl_clock.py --schedule 3 --batches 1048576 2>&1 | grep -v amdgpu.ids >> $out/clock.txt
RAYEN_HIP_LIBRARY=$V/librayen_mfma_pair_wl_$v.so timeout 300 python scripts/ubench/wl_clock.py --schedule 3 --batches 524288 --reserve 128 2>&1 | grep -v amdgpu.ids >> $out/clock.txt
done
cat $out/clock.txt
