#!/bin/bash
# run bench.py against variant libraries:  ablate_run.sh "<bench args>" variant...
cd "$(dirname "$0")/../.."
args="$1"; shift
for v in "$@"; do
  RAYEN_HIP_LIBRARY=$PWD/scripts/ubench/variants/librayen_$v.so python bench.py --no-cpu-baseline $args 2>/dev/null |
    python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', '$args', '%.4f ms' % d['ms_per_step'], '%.3e' % d['value'], 'frac %.3f' % d['roofline']['frac'])"
done
