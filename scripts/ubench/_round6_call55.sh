out=gpurun_out/r06zzk; mkdir -p $out
timeout 2400 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider -x > $out/pytest_gpu.log 2>&1; tail -5 $out/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
