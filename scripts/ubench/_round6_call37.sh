out=gpurun_out/r06zs; mkdir -p $out
for lib in default iofabl1 iofabl2 iofabl3; do
  if [ $lib = default ]; then unset RAYEN_HIP_LIBRARY; else export RAYEN_HIP_LIBRARY=$PWD/scripts/ubench/variants/librayen_mfma_pair_io_$lib.so; fi
  timeout 300 python scripts/ubench/io_bench.py --config c5 --batches 262144,1048576 2>&1 | grep -v amdgpu | tail -1 >> $out/iof_abl.txt
done
cat $out/iof_abl.txt
