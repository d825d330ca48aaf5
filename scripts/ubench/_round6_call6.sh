set -x
out=gpurun_out/r06f; mkdir -p $out
for i in 1 2; do
  RAYEN_LMI_QUAD_MM=0 timeout 300 python bench.py --config c4 --no-cpu-baseline --no-families > $out/bench_c4_mm0_$i.json 2>$out/err.txt
  RAYEN_LMI_QUAD_MM=1 timeout 300 python bench.py --config c4 --no-cpu-baseline --no-families > $out/bench_c4_mm1_$i.json 2>$out/err.txt
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r06f/bench_c4*.json')):
    try:
        d=json.load(open(f)); print(f.split('/')[-1], 'ms', round(d['ms_per_step'],5), 'kernel_ms', round(d['kernel_ms'],5), 'viol', d['max_violation'])
    except Exception as e: print(f, 'ERR', e)
PY
timeout 1500 python -m pytest tests/test_gpu_lmi_wave.py tests/test_gpu_parity.py -m gpu -q --timeout 900 -p no:cacheprovider --maxfail=10 -k "c4 or lmi or golden or quad" > $out/pytest.log 2>&1; tail -6 $out/pytest.log
