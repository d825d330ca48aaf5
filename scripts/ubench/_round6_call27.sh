out=gpurun_out/r06zg; mkdir -p $out
timeout 1200 python -m pytest tests/test_gpu_pair_wl.py -m gpu -q -x --timeout 900 -p no:cacheprovider > $out/pytest_wl.log 2>&1; tail -5 $out/pytest_wl.log
timeout 600 python bench.py --steps 20 --warmup 5 > $out/bench_c3.json 2> $out/bench_c3.err; python - <<'PY'
import json
d=json.load(open('gpurun_out/r06zg/bench_c3.json'))
print({k:d[k] for k in ('value','ms_per_step','kernel_ms','max_violation') if k in d}, d['config'].get('kernel'))
print('roofline', {k:v for k,v in d['roofline'].items() if k in ('frac','achieved','hbm_GBps','hbm_frac','l3_resident')})
print('io leg', d.get('pair_kernel_with_trickled_rows',{}).get('ms_per_step'))
PY
timeout 1500 python -m pytest tests/test_gpu_pair_io.py tests/test_gpu_pair_ws.py -m gpu -q -x --timeout 900 -p no:cacheprovider > $out/pytest_sched.log 2>&1; tail -4 $out/pytest_sched.log
