# round 4, last call: smoke(), then the self-consistency stress at ten times the launches on the sets with an NA_E write-out
out=gpurun_out/r04v; mkdir -p $out
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -4 | tee $out/smoke.txt
timeout 2400 python scripts/ubench/determinism_stress.py --reps 3000 --configs c5,c5r 2>&1 | grep "^{" | tee $out/determinism_c5.txt
