# round 4, session 5: every kernel family against itself under repetition
out=gpurun_out/r04u; mkdir -p $out
timeout 2400 python scripts/ubench/determinism_stress.py --reps 300 2>&1 | grep "^{" | tee $out/determinism.txt
