#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_lmi_mixed.py -m gpu -x -q -k "old_head" 2>&1 | tail -25 | cut -c1-220
