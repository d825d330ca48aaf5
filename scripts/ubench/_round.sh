mkdir -p gpurun_out/r02m
timeout 2400 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider > gpurun_out/r02m/pytest.log 2>&1
tail -4 gpurun_out/r02m/pytest.log
timeout 300 python bench.py --force-dist --gather --no-cpu-baseline > gpurun_out/r02m/bench_c3_rccl1_gather.json 2> gpurun_out/r02m/bench_c3_rccl1_gather.err
tail -c 700 gpurun_out/r02m/bench_c3_rccl1_gather.json; tail -3 gpurun_out/r02m/bench_c3_rccl1_gather.err
timeout 300 python bench.py --force-dist --gather --no-cpu-baseline --config c5 --scaling strong --batch 262144 --chunks 4 > gpurun_out/r02m/bench_c5_rccl1_gather.json 2>&1
tail -c 400 gpurun_out/r02m/bench_c5_rccl1_gather.json
