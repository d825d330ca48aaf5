#!/bin/bash
out=gpurun_out/r05m; mkdir -p $out
export TMPDIR=/tmp
U=$PWD/scripts/ubench/variants
line() { python - "$1" "$2" <<'P'
import json,sys
try:
    d=json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    print(sys.argv[1], round(d["ms_per_step"],5), d["config"].get("kernel","")[:36])
except Exception as e:
    print(sys.argv[1], "failed", e)
P
}
timeout 900 python -m pytest tests/test_gpu_mapper.py tests/test_gpu_pair_io.py tests/test_gpu_pair_ws.py tests/test_gpu_boundary.py -m gpu -x -q --timeout 600 -p no:cacheprovider > $out/pytest_sel.log 2>&1; echo "rc=$?" >> $out/pytest_sel.log
tail -3 $out/pytest_sel.log
for rep in 1 2; do
RAYEN_HIP_LIBRARY=$U/librayen_r04.so timeout 400 python bench.py --mapper 64 --no-cpu-baseline --no-families > $out/bench_map_r04.json 2>/dev/null; line r04_map64 $out/bench_map_r04.json
timeout 400 python bench.py --mapper 64 --no-cpu-baseline --no-families > $out/bench_map_new.json 2>/dev/null; line new_map64 $out/bench_map_new.json
done
RAYEN_HIP_LIBRARY=$U/librayen_r04.so timeout 400 python bench.py --config c5 --mapper 32 --no-cpu-baseline --no-families > $out/bench_map5_r04.json 2>/dev/null; line r04_c5map32 $out/bench_map5_r04.json
timeout 400 python bench.py --config c5 --mapper 32 --no-cpu-baseline --no-families > $out/bench_map5_new.json 2>/dev/null; line new_c5map32 $out/bench_map5_new.json
