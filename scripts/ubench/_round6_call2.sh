set -x
out=gpurun_out/r06b; mkdir -p $out
timeout 1200 python -m pytest tests/test_gpu_pair_io.py tests/test_gpu_feasibility.py -m gpu -q -x --timeout 600 -p no:cacheprovider -s > $out/pytest_a.log 2>&1; tail -3 $out/pytest_a.log; grep "rows > 0" $out/pytest_a.log
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 600 -p no:cacheprovider -k "nan or NaN or c3 or golden or fp32_bar" > $out/pytest_b.log 2>&1; tail -3 $out/pytest_b.log
for i in 1 2; do
  timeout 300 python bench.py --no-cpu-baseline --no-families > $out/bench_c3_new_$i.json 2>/dev/null
  RAYEN_HIP_LIBRARY=$PWD/scripts/ubench/variants/librayen_oldio.so timeout 300 python bench.py --no-cpu-baseline --no-families > $out/bench_c3_old_$i.json 2>/dev/null
  timeout 300 python bench.py --config c5 --no-cpu-baseline --no-families > $out/bench_c5_new_$i.json 2>/dev/null
  RAYEN_HIP_LIBRARY=$PWD/scripts/ubench/variants/librayen_oldio.so timeout 300 python bench.py --config c5 --no-cpu-baseline --no-families > $out/bench_c5_old_$i.json 2>/dev/null
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r06b/bench_*.json')):
    try:
        d=json.load(open(f)); r=d['roofline']
        print(f.split('/')[-1], 'ms', round(d['ms_per_step'],5), 'kernel_ms', round(d['kernel_ms'],5), 'l3', r.get('l3_resident',{}).get('kernel_ms'), 'pairs', r['footprint']['pairs_in_rotation'], 'viol', d['max_violation'])
    except Exception as e: print(f, 'ERR', e)
PY
