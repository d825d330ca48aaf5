#!/bin/bash
# run bench.py against each variant library at a few batch sizes
cd "$(dirname "$0")/../.."
for v in "$@"; do
  for B in 262144 524288 1048576; do
    RAYEN_HIP_LIBRARY=$PWD/scripts/ubench/variants/librayen_$v.so python bench.py --no-cpu-baseline --batch $B 2>/dev/null |
      python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', $B, '%.4f ms' % d['ms_per_step'], '%.3e' % d['value'], 'frac %.3f' % d['roofline']['frac'])"
  done
done
