out=gpurun_out/r06zx2; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_pair_wl.py -m gpu -q -x --timeout 600 -p no:cacheprovider > $out/pytest_wl.log 2>&1; tail -4 $out/pytest_wl.log
for i in 1 2; do timeout 200 python scripts/ubench/wl_check.py 2>&1 | grep -v amdgpu | grep "time us" ; done
