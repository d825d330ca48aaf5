out=gpurun_out/r06w; mkdir -p $out
V=$PWD/scripts/ubench/variants
for lib in clock clock3; do
export RAYEN_HIP_LIBRARY=$V/librayen_mfma_pair_wl_$lib.so
echo "== $lib" >> $out/clock.txt
timeout 200 python scripts/ubench/wl_clock.py --schedule 3 --batches 1048576 2>&1 | grep -v amdgpu.ids >> $out/clock.txt
timeout 200 python scripts/ubench/wl_clock.py --schedule 3 --batches 524288 --reserve 128 2>&1 | grep -v amdgpu.ids >> $out/clock.txt
timeout 200 python scripts/ubench/wl_clock.py --schedule 3 --batches 262144 --reserve 192 2>&1 | grep -v amdgpu.ids >> $out/clock.txt
done
cat $out/clock.txt
