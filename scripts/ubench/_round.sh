#!/bin/bash
mkdir -p gpurun_out/r05zj
o=gpurun_out/r05zj/lmi_wpe_ab.txt; : > $o
S="40x10 64x10 80x10 100x10 100x100 128x100 140x100"
timeout 400 python scripts/ubench/lmi_bwd_ab.py $S 2>&1 | grep -v amdgpu.ids >> $o
for v in wpe6 wpe8; do RAYEN_HIP_LIBRARY=scripts/ubench/variants/librayen_lmi_block_$v.so timeout 400 python scripts/ubench/lmi_bwd_ab.py $S 2>&1 | grep -v amdgpu.ids | sed "s/^{/{\"lib\": \"$v\", /" >> $o; done
cat $o
