#!/bin/bash
# Developer timing of the W-stationary kernel and its ablation builds (wrong results) on config 3:
#   scripts/ubench/ws_bench.sh [bits ...]        (run on the GPU box; builds the variants first if hipcc is there)
REPO="$(cd "$(dirname "$0")/../.." && pwd)"
cd "$REPO"
export PYTHONPATH=$REPO
python scripts/ubench/io_bench.py --schedule 2 --batches 32768,131072,262144,1048576 2>&1 | grep -v amdgpu.ids
python scripts/ubench/io_bench.py --schedule 1 --batches 131072,262144,1048576 2>&1 | grep -v amdgpu.ids
for lib in scripts/ubench/variants/librayen_mfma_pair_ws*.so; do
  [ -f "$lib" ] || continue
  RAYEN_HIP_LIBRARY=$lib python scripts/ubench/io_bench.py --schedule 2 --batches 131072,262144,1048576 2>&1 | grep -v amdgpu.ids
done
