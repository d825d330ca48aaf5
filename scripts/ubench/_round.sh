#!/bin/bash
mkdir -p gpurun_out/r05ze
o=gpurun_out/r05ze/lmi_256_ab.txt; : > $o
for m in 0 129; do RAYEN_HIP_LIBRARY=scripts/ubench/variants/librayen_lmi_block_t256.so RAYEN_LB_256_UPTO=$m timeout 300 python scripts/ubench/lmi_bwd_ab.py 2>&1 | grep -v amdgpu.ids >> $o; done
cat $o
