out=gpurun_out/r06zd; mkdir -p $out
V=$PWD/scripts/ubench/variants
timeout 200 python scripts/ubench/io_bench.py --schedule 3 --batches 262144,1048576 2>&1 | grep -v amdgpu.ids | sed "s/^/base: /" >> $out/prio.txt
for v in prio1 prio2 prio3 prio4; do
RAYEN_HIP_LIBRARY=$V/librayen_mfma_pair_wl_$v.so timeout 200 python scripts/ubench/io_bench.py --schedule 3 --batches 262144,1048576 2>&1 | grep -v amdgpu.ids | sed "s/^/$v: /" >> $out/prio.txt
done
timeout 200 python scripts/ubench/io_bench.py --schedule 3 --batches 262144,1048576 2>&1 | grep -v amdgpu.ids | sed "s/^/base: /" >> $out/prio.txt
cat $out/prio.txt
