#!/bin/bash
mkdir -p gpurun_out/r05zn
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > gpurun_out/r05zn/pytest_full.log
cat gpurun_out/r05zn/pytest_full.log
