out=gpurun_out/r06zq; mkdir -p $out
V=$PWD/scripts/ubench/variants
echo "== base8" > $out/bwd_bench.txt; timeout 300 python scripts/ubench/bwd_bench.py c3 2>&1 | grep -v amdgpu >> $out/bwd_bench.txt
for v in l8 l12; do echo "== $v" >> $out/bwd_bench.txt; RAYEN_HIP_LIBRARY=$V/librayen_mfma_bwdd_$v.so timeout 300 python scripts/ubench/bwd_bench.py c3 2>&1 | grep -v amdgpu >> $out/bwd_bench.txt; done
echo "== old" >> $out/bwd_bench.txt; RAYEN_BWD_DENSE_PAIRS=0 timeout 300 python scripts/ubench/bwd_bench.py c3 2>&1 | grep -v amdgpu >> $out/bwd_bench.txt
cat $out/bwd_bench.txt
