out=gpurun_out/r06zzd; mkdir -p $out
B=4096,8192,16384,32768,49152,65536,81920,98304,131072
echo "== others (default thresholds: wl from 98304)" >> $out/sweep.txt
timeout 300 python scripts/ubench/io_bench.py --config c3 --batches $B 2>&1 | grep -v amdgpu | tail -1 >> $out/sweep.txt
echo "== wl forced from 1 group" >> $out/sweep.txt
RAYEN_WL_MIN_GROUPS=1 timeout 300 python scripts/ubench/io_bench.py --config c3 --batches $B 2>&1 | grep -v amdgpu | tail -1 >> $out/sweep.txt
echo "== others again" >> $out/sweep.txt
RAYEN_WL_MIN_GROUPS=100000000 timeout 300 python scripts/ubench/io_bench.py --config c3 --batches $B 2>&1 | grep -v amdgpu | tail -1 >> $out/sweep.txt
echo "== wl again" >> $out/sweep.txt
RAYEN_WL_MIN_GROUPS=1 timeout 300 python scripts/ubench/io_bench.py --config c3 --batches $B 2>&1 | grep -v amdgpu | tail -1 >> $out/sweep.txt
cat $out/sweep.txt
