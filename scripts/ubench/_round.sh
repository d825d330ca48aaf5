#!/bin/bash
mkdir -p gpurun_out/r05zr
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > gpurun_out/r05zr/pytest_full.log
cat gpurun_out/r05zr/pytest_full.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 600 python scripts/ubench/lmi_sweep.py > gpurun_out/r05zr/lmi_sweep.txt 2>&1
cut -c1-200 gpurun_out/r05zr/lmi_sweep.txt
