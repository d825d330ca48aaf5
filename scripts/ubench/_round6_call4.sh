set -x
out=gpurun_out/r06d; mkdir -p $out
timeout 2400 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider -x --maxfail=15 > $out/pytest.log 2>&1
echo "pytest rc=$?" >> $out/pytest.log
tail -30 $out/pytest.log
