#!/bin/bash
out=gpurun_out/r05k; mkdir -p $out
export TMPDIR=/tmp
U=$PWD/scripts/ubench/variants
line() { python - "$1" "$2" <<'P'
import json,sys
try:
    d=json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    ts=d.get("training_step",{})
    print(sys.argv[1], round(d["ms_per_step"],5), d["config"].get("kernel","")[:36], "| viol", d.get("max_violation"), "| train fwd/bwd", ts.get("forward_with_record_ms"), ts.get("backward_ms"), "| exact", (d.get("families") or {}).get("exact_fp32_ms"))
except Exception as e:
    print(sys.argv[1], "failed", e)
P
}
for cfg in c3 c1 c2 c4 c5 c5r; do
  RAYEN_HIP_LIBRARY=$U/librayen_r04.so timeout 400 python bench.py --config $cfg --no-cpu-baseline > $out/bench_${cfg}_r04.json 2> $out/bench_${cfg}_r04.err; line r04_$cfg $out/bench_${cfg}_r04.json
  timeout 400 python bench.py --config $cfg --no-cpu-baseline > $out/bench_${cfg}_new.json 2> $out/bench_${cfg}_new.err; line new_$cfg $out/bench_${cfg}_new.json
done
RAYEN_HIP_LIBRARY=$U/librayen_r04.so timeout 400 python bench.py --mapper 64 --no-cpu-baseline > $out/bench_map_r04.json 2>/dev/null; line r04_map64 $out/bench_map_r04.json
timeout 400 python bench.py --mapper 64 --no-cpu-baseline > $out/bench_map_new.json 2>/dev/null; line new_map64 $out/bench_map_new.json
timeout 2400 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider -x > $out/pytest_full.log 2>&1; echo "rc=$?" >> $out/pytest_full.log
tail -4 $out/pytest_full.log
