set -x
out=gpurun_out/r06e; mkdir -p $out
for i in 1 2; do
  RAYEN_BWD_PAIR=0 timeout 300 python scripts/ubench/bwd_pair_ab.py >> $out/bwd_pair_ab.txt 2>$out/err0.txt
  RAYEN_BWD_PAIR=1 timeout 300 python scripts/ubench/bwd_pair_ab.py >> $out/bwd_pair_ab.txt 2>$out/err1.txt
done
timeout 300 python scripts/ubench/bwd_pair_ab.py >> $out/bwd_pair_ab.txt 2>$out/err2.txt
cat $out/bwd_pair_ab.txt; tail -3 $out/err1.txt
timeout 1500 python -m pytest tests/test_gpu_backward.py tests/test_gpu_backward_pairs.py tests/test_gpu_feasibility.py -m gpu -q --timeout 900 -p no:cacheprovider --maxfail=10 > $out/pytest.log 2>&1; tail -8 $out/pytest.log
