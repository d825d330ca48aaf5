#!/bin/bash
# scratch: one gpurun call
mkdir -p gpurun_out/r05y
timeout 1500 python -m pytest tests/test_gpu_lmi_mixed.py -m gpu -q 2>&1 | tail -60 > gpurun_out/r05y/pytest_mixed.log
cat gpurun_out/r05y/pytest_mixed.log | cut -c1-250
