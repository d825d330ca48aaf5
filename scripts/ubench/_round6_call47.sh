out=gpurun_out/r06zzc; mkdir -p $out
for lib in default stag8 stag24 stag48 default stag8 stag24 stag48; do
  if [ $lib = default ]; then unset RAYEN_HIP_LIBRARY; else export RAYEN_HIP_LIBRARY=$PWD/scripts/ubench/variants/librayen_mfma_pair_wl_$lib.so; fi
  echo "== $lib" >> $out/stag.txt
  timeout 200 python scripts/ubench/wl_check.py --batches 262144,524288 2>&1 | grep "time us" >> $out/stag.txt
done
cat $out/stag.txt
