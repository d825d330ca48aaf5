#!/bin/bash
out=gpurun_out/r05n; mkdir -p $out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_lmi_wave.py -m gpu -x -q --timeout 900 -p no:cacheprovider > $out/pytest_lmi.log 2>&1; echo "rc=$?" >> $out/pytest_lmi.log
tail -15 $out/pytest_lmi.log
timeout 900 python scripts/ubench/lmi_block_bench.py > $out/lmi_block_bench.txt 2>&1; cat $out/lmi_block_bench.txt | cut -c1-250
