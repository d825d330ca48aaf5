#!/bin/bash
# Developer helper: rebuild ONE translation unit with extra -D flags and link it against the objects of the last regular
# build:  scripts/ubench/tu_variant.sh <tu (e.g. rayen_mfma_pair)> <name> [-D...]
#   ->  scripts/ubench/variants/librayen_<tu>_<name>.so   (run against it with RAYEN_HIP_LIBRARY=<that path>)
REPO="$(cd "$(dirname "$0")/../.." && pwd)"
tu="$1"; name="$2"; shift 2
out="$REPO/scripts/ubench/variants/lib${tu}_$name.so"
mkdir -p "$(dirname "$out")"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I "$REPO/include" -I "$REPO/rayen_amd/csrc" "$@" \
  -c "$REPO/rayen_amd/csrc/$tu.hip" -o /tmp/tu_variant_${tu}_$name.o || exit 1
objs=$(ls "$REPO"/rayen_amd/csrc/_obj/*.o | grep -v "/$tu.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs /tmp/tu_variant_${tu}_$name.o -o "$out" && echo "$out"
