out=gpurun_out/r06zzi; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_pair_wl.py -m gpu -q -x --timeout 600 -p no:cacheprovider > $out/pytest_wl.log 2>&1; tail -4 $out/pytest_wl.log
timeout 300 python bench.py --config c2 --no-cpu-baseline > $out/bench_c2.json 2> $out/bench_c2.err; python -c "
import json;d=json.loads(open('$out/bench_c2.json').read().strip().splitlines()[-1]);print(d['value'], d['ms_per_step'], d['config'].get('kernel'), d['roofline'].get('kernel_ms'))"
RAYEN_WL_MIN_GROUPS=100000000 timeout 300 python bench.py --config c2 --no-cpu-baseline > $out/bench_c2_old.json 2> $out/bench_c2_old.err; python -c "
import json;d=json.loads(open('$out/bench_c2_old.json').read().strip().splitlines()[-1]);print(d['value'], d['ms_per_step'], d['config'].get('kernel'), d['roofline'].get('kernel_ms'))"
