# round 4, closing check of the library as it lies in the tree: smoke(), the schedule-equality file (with the many-launch tests), the boundary file
out=gpurun_out/r04z; mkdir -p $out
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $out/smoke.txt
timeout 1200 python -m pytest tests/test_gpu_pair_io.py tests/test_gpu_boundary.py -m gpu -q --timeout 600 -p no:cacheprovider 2>&1 | tail -3 | tee $out/pytest.txt
