out=gpurun_out/r06zze; mkdir -p $out
B=32,256,512,1024,2048,4096
for cfg in c3; do
echo "== $cfg others" >> $out/sweep.txt
RAYEN_WL_MIN_GROUPS=100000000 timeout 300 python scripts/ubench/io_bench.py --config $cfg --batches $B 2>&1 | grep -v amdgpu | tail -1 >> $out/sweep.txt
echo "== $cfg wl forced from 1 group" >> $out/sweep.txt
RAYEN_WL_MIN_GROUPS=1 timeout 300 python scripts/ubench/io_bench.py --config $cfg --batches $B 2>&1 | grep -v amdgpu | tail -1 >> $out/sweep.txt
done
cat $out/sweep.txt
