# round 4, last session: the ten fp32 LMI-backward seeds of the 1000-seed fuzz on variants of the inverse iteration (iterations / shift)
out=gpurun_out/r04y; mkdir -p $out
ids=""; for s in 71 74 78 111 133 150 189 216 229 234; do ids="$ids tests/test_gpu_backward.py::test_random_lmi_sets_backward[dtype0-$s]"; done
for lib in rayen_amd/csrc/librayen_hip.so scripts/ubench/variants/librayen_lmi_it5.so scripts/ubench/variants/librayen_lmi_it5s.so scripts/ubench/variants/librayen_lmi_s1.so; do
  RAYEN_HIP_LIBRARY=$PWD/$lib RAYEN_FUZZ_SEEDS=1000 timeout 600 python -m pytest $ids -m gpu -q --timeout 300 -p no:cacheprovider 2>&1 | grep -E "AssertionError: \(|passed|failed" | cut -c1-260 | sed "s|^|$(basename $lib) |"
done 2>&1 | tee $out/lmi_seeds.txt
