set -x
mkdir -p gpurun_out/r06a
./scripts/ubench/valu_issue > gpurun_out/r06a/valu_issue.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_lmi_mixed.py tests/test_abi_load.py tests/test_gpu_boundary.py -m gpu -q -x --timeout 600 -p no:cacheprovider > gpurun_out/r06a/pytest_a.log 2>&1; tail -3 gpurun_out/r06a/pytest_a.log
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 600 -p no:cacheprovider -k "c5 or golden" -s > gpurun_out/r06a/pytest_b.log 2>&1; tail -3 gpurun_out/r06a/pytest_b.log; grep "oracle NaN" gpurun_out/r06a/pytest_b.log | head -20
timeout 600 python scripts/ubench/inward_bias.py > gpurun_out/r06a/inward_bias.txt 2>&1; cat gpurun_out/r06a/inward_bias.txt
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r06a/bench_c3_driver_like.json 2> gpurun_out/r06a/bench_c3.err; head -c 1500 gpurun_out/r06a/bench_c3_driver_like.json
