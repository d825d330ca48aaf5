#!/bin/bash
mkdir -p gpurun_out/r05zu
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > gpurun_out/r05zu/pytest_full.log
cat gpurun_out/r05zu/pytest_full.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | cut -c1-300
