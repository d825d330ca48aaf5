#!/bin/bash
# Developer helper: build librayen_hip.so with extra -D flags into scripts/ubench/variants/librayen_<name>.so
# (run a bench against it with RAYEN_HIP_LIBRARY=<that path>).   scripts/build_variant.sh noload -DRAYEN_ABL_NOLOAD
REPO="$(cd "$(dirname "$0")/.." && pwd)"
name="$1"; shift
out="$REPO/scripts/ubench/variants/librayen_$name.so"
mkdir -p "$(dirname "$out")"
cd "$REPO/rayen_amd/csrc" || exit 1
pids=()
for f in rayen_abi rayen_generic rayen_mfma rayen_mfma_split rayen_mfma_mapped rayen_mfma_bwd rayen_mfma_bwdg rayen_mfma_bwdg64 rayen_mfma_bwd64 rayen_mfma_f64 rayen_lmi_quad32 rayen_lmi_quad64; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I "$REPO/include" -I . "$@" -c $f.hip -o /tmp/variant_${name}_$f.o &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p || exit 1; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC /tmp/variant_${name}_*.o -o "$out" && echo "$out"
