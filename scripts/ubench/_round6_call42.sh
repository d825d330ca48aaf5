out=gpurun_out/r06zw; mkdir -p $out
timeout 1200 python -m pytest tests/test_gpu_pair_wl.py -m gpu -q -x --timeout 900 -p no:cacheprovider > $out/pytest_wl.log 2>&1; tail -3 $out/pytest_wl.log
timeout 300 python scripts/ubench/wl_check.py --batches 98304,262144,1048576 2>&1 | grep -v amdgpu.ids > $out/wl_check.txt; cat $out/wl_check.txt
for i in 1 2; do timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-families > $out/bench_c3_$i.json 2> $out/bench_c3.err; done
python - <<PY
import json
for i in (1,2):
    d=json.load(open("gpurun_out/r06zw/bench_c3_%d.json"%i))
    print({k:d[k] for k in ("value","ms_per_step","kernel_ms") if k in d}, d["config"].get("kernel")[:24], "l3", d["roofline"].get("l3_resident",{}).get("kernel_ms"), d.get("training_step",{}).get("backward_ms"), d.get("training_step",{}).get("forward_with_record_ms"))
PY
