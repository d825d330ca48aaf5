#!/bin/bash
# Developer helper: rebuild ONLY rayen_mfma_pair_io.hip with extra -D flags and link it against the objects of the last
# regular build:  scripts/ubench/io_variant.sh <name> [-D...]  ->  scripts/ubench/variants/librayen_io_<name>.so
# (run against it with RAYEN_HIP_LIBRARY=<that path>)
REPO="$(cd "$(dirname "$0")/../.." && pwd)"
name="$1"; shift
out="$REPO/scripts/ubench/variants/librayen_io_$name.so"
mkdir -p "$(dirname "$out")"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I "$REPO/include" -I "$REPO/rayen_amd/csrc" "$@" \
  -c "$REPO/rayen_amd/csrc/rayen_mfma_pair_io.hip" -o /tmp/io_variant_$name.o || exit 1
objs=$(ls "$REPO"/rayen_amd/csrc/_obj/*.o | grep -v rayen_mfma_pair_io.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs /tmp/io_variant_$name.o -o "$out" && echo "$out"
