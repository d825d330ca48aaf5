#!/bin/bash
# scratch: one gpurun call
mkdir -p gpurun_out/r05w
timeout 1500 python -m pytest tests/test_gpu_lmi_wave.py -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r05w/pytest_lmi.log
cat gpurun_out/r05w/pytest_lmi.log
timeout 600 python scripts/ubench/lmi_sweep.py > gpurun_out/r05w/lmi_sweep.txt 2>&1
cut -c1-200 gpurun_out/r05w/lmi_sweep.txt
timeout 600 python scripts/ubench/lmi_block_bench.py > gpurun_out/r05w/lmi_block_bench.txt 2>&1
cut -c1-250 gpurun_out/r05w/lmi_block_bench.txt
