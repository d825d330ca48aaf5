out=gpurun_out/r06zzf; mkdir -p $out
timeout 2300 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider -x > $out/pytest_gpu.log 2>&1; tail -6 $out/pytest_gpu.log
