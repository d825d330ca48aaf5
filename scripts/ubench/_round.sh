#!/bin/bash
timeout 1500 python -m pytest tests/test_gpu_lmi_mixed.py tests/test_gpu_lmi_wave.py -m gpu -x -q 2>&1 | tail -12 | cut -c1-250
