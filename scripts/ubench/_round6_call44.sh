out=gpurun_out/r06zza; mkdir -p $out
for lib in head0 head1 head2 head0 head1 head2; do
  export RAYEN_HIP_LIBRARY=$PWD/scripts/ubench/variants/librayen_mfma_pair_wl_$lib.so
  echo "== $lib" >> $out/head.txt
  timeout 200 python scripts/ubench/wl_check.py --batches 262144,1048576 2>&1 | grep "time us" >> $out/head.txt
done
cat $out/head.txt
