#!/bin/bash
mkdir -p gpurun_out/r05zt
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > gpurun_out/r05zt/pytest_full.log
cat gpurun_out/r05zt/pytest_full.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
