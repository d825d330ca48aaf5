#!/bin/bash
# Register / scratch / LDS use of every kernel in one translation unit:  scripts/kernel_resources.sh rayen_mfma
REPO="$(cd "$(dirname "$0")/.." && pwd)"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I "$REPO/include" -I "$REPO/rayen_amd/csrc" \
  -c "$REPO/rayen_amd/csrc/$1.hip" -o /tmp/$1.resources.o -Rpass-analysis=kernel-resource-usage 2>&1 |
python3 -c "
import re, subprocess, sys
cur, vals = None, {}
for line in sys.stdin:
    m = re.search(r'remark:\s+([\w \[\]/]+?):\s+(\S+)', line)
    if not m:
        continue
    key, val = m.group(1).strip(), m.group(2)
    if key == 'Function Name':
        cur, vals = val, {}
        continue
    vals[key] = val
    if key.startswith('LDS Size') and cur:
        name = subprocess.run(['c++filt', cur], capture_output=True, text=True).stdout.strip()
        name = re.sub(r'^void ', '', re.sub(r'\(.*', '', name))
        print(f\"{name:52s} VGPR {vals.get('VGPRs','?'):>4} AGPR {vals.get('AGPRs','?'):>4} vspill {vals.get('VGPRs Spill','?'):>3} sspill {vals.get('SGPRs Spill','?'):>3} scratch {vals.get('ScratchSize [bytes/lane]','?'):>4} occ {vals.get('Occupancy [waves/SIMD]','?')} LDS {val}\")
"
