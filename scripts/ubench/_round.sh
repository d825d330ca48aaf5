#!/bin/bash
mkdir -p gpurun_out/r05zd
timeout 1500 python -m pytest tests/test_gpu_lmi_wave.py tests/test_gpu_lmi_mixed.py -m gpu -x -q 2>&1 | tail -5
o=gpurun_out/r05zd/lmi_bwd_ab.txt; : > $o
timeout 300 python scripts/ubench/lmi_bwd_ab.py 2>&1 | grep -v amdgpu.ids >> $o
cat $o
