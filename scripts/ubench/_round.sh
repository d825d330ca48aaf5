#!/bin/bash
# scratch: one gpurun call
mkdir -p gpurun_out/r05zi
timeout 2400 python -m pytest tests -m gpu -x -q --durations=12 2>&1 | tail -30 > gpurun_out/r05zi/pytest_full.log
cat gpurun_out/r05zi/pytest_full.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -4 > gpurun_out/r05zi/smoke.log
cat gpurun_out/r05zi/smoke.log
timeout 600 python scripts/ubench/lmi_sweep.py > gpurun_out/r05zi/lmi_sweep.txt 2>&1
cut -c1-200 gpurun_out/r05zi/lmi_sweep.txt
timeout 600 python scripts/ubench/lmi_block_bench.py > gpurun_out/r05zi/lmi_block_bench.txt 2>&1
cut -c1-200 gpurun_out/r05zi/lmi_block_bench.txt
