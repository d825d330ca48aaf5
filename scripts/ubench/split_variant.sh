#!/bin/bash
# Developer helper: rebuild only rayen_mfma_split.hip with extra flags and link it against the objects of the last
# regular build -> scripts/ubench/variants/librayen_<name>.so     split_variant.sh nt1 -DSOME_EXPERIMENT=1
REPO="$(cd "$(dirname "$0")/../.." && pwd)"
name="$1"; shift
out="$REPO/scripts/ubench/variants/librayen_$name.so"
mkdir -p "$(dirname "$out")"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I "$REPO/include" -I "$REPO/rayen_amd/csrc" "$@" \
  -c "$REPO/rayen_amd/csrc/rayen_mfma_split.hip" -o /tmp/splitvar_$name.o || exit 1
objs=$(ls "$REPO"/rayen_amd/csrc/_obj/*.o | grep -v rayen_mfma_split)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs /tmp/splitvar_$name.o -o "$out" && echo "$out"
