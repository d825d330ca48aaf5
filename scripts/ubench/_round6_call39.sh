out=gpurun_out/r06zu; mkdir -p $out
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_c3.json 2> $out/bench_c3.err; tail -c 3000 $out/bench_c3.json
