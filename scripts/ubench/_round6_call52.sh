out=gpurun_out/r06zzh; mkdir -p $out
for B in 1024 4096 16384 32768 65536 131072; do
  for m in 100000000 1; do
    echo "B=$B RAYEN_BWDD_MIN_GROUPS=$m $(BWD_B=$B RAYEN_BWDD_MIN_GROUPS=$m timeout 200 python scripts/ubench/bwd_bench.py c3 2>&1 | grep -v amdgpu | tail -1 | cut -c1-170)" >> $out/bwdd_small.txt
  done
done
cat $out/bwdd_small.txt
