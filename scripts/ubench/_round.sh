#!/bin/bash
# scratch: one gpurun call
mkdir -p gpurun_out/r05s
RAYEN_HIP_LIBRARY=scripts/ubench/variants/librayen_lmi_block_prof.so timeout 600 python scripts/ubench/lmi_block_prof.py > gpurun_out/r05s/lmi_block_prof.txt 2>&1
cat gpurun_out/r05s/lmi_block_prof.txt
