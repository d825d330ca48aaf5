#!/bin/bash
mkdir -p gpurun_out/r05zl
timeout 1500 python -m pytest tests/test_gpu_lmi_wave.py tests/test_gpu_lmi_mixed.py -m gpu -x -q 2>&1 | tail -8
timeout 400 python scripts/ubench/lmi_bwd_ab.py 150x100 180x100 196x100 220x100 250x100 280x10 300x100 2>&1 | grep -v amdgpu.ids > gpurun_out/r05zl/wp.txt
LMI_DTYPE=f64 timeout 400 python scripts/ubench/lmi_bwd_ab.py 150x10 196x10 210x10 2>&1 | grep -v amdgpu.ids >> gpurun_out/r05zl/wp.txt
cat gpurun_out/r05zl/wp.txt
