#!/bin/bash
# scratch script of the current gpurun call (rewritten per call)
out=gpurun_out/r05e; mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_pair_io.py tests/test_gpu_pair_ws.py -m gpu -x -q --timeout 600 -p no:cacheprovider > $out/pytest_pair.log 2>&1; echo "rc=$?" >> $out/pytest_pair.log
tail -3 $out/pytest_pair.log
run() {  # tag, env...
  tag=$1; shift
  env "$@" timeout 300 python bench.py --config c3 --no-cpu-baseline --no-families > $out/bench_$tag.json 2> $out/bench_$tag.err
  python - <<P
import json
try:
    d=json.loads(open("$out/bench_$tag.json").read().strip().splitlines()[-1])
    print("$tag", round(d["ms_per_step"],5), round(d.get("kernel_ms",0),5), d["config"]["kernel"][:30])
except Exception as e:
    print("$tag failed", e)
P
}
for rep in 1 2; do
  run r04_$rep RAYEN_HIP_LIBRARY=$PWD/scripts/ubench/variants/librayen_r04.so
  run tri1_$rep RAYEN_PAIR_TRI=1
  run tri0_$rep RAYEN_PAIR_TRI=0
done
