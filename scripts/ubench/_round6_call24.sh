out=gpurun_out/r06za; mkdir -p $out
RAYEN_WL_MIN_PER_WAVE=0 timeout 600 python scripts/ubench/wl_check.py --batches 8192,32768,65536,98304,131072,196608,262144,393216,524288 2>&1 | grep -v amdgpu.ids > $out/wl_check.txt
cat $out/wl_check.txt
