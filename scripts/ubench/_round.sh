#!/bin/bash
out=gpurun_out/r05p; mkdir -p $out
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider > $out/pytest_full.log 2>&1; echo "rc=$?" >> $out/pytest_full.log
tail -6 $out/pytest_full.log
python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; tail -2 $out/smoke.log
timeout 600 python scripts/ubench/lmi_sweep.py 2>&1 | grep "^{" > $out/lmi_sweep.txt; cat $out/lmi_sweep.txt | cut -c1-220
