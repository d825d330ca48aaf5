out=gpurun_out/r06zzj; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_pair_wl.py -m gpu -q -x --timeout 600 -p no:cacheprovider -k "mapped" > $out/pytest_wlm.log 2>&1; tail -15 $out/pytest_wlm.log
timeout 1200 python -m pytest tests/test_gpu_mapper.py -m gpu -q -x --timeout 600 -p no:cacheprovider > $out/pytest_mapper.log 2>&1; tail -5 $out/pytest_mapper.log
timeout 300 python bench.py --mapper 64 --no-cpu-baseline --no-families > $out/bench_map.json 2> $out/bench_map.err; python -c "
import json;d=json.loads(open('$out/bench_map.json').read().strip().splitlines()[-1]);print(d['value'], d['ms_per_step'], d['config'].get('kernel'), d['roofline'].get('kernel_ms'))"
