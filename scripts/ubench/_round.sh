# round 4, session 5: stress of every trickled-row instance on the fixed build, and of the sets not yet stressed on the base build
out=gpurun_out/r04t; mkdir -p $out
timeout 1200 python scripts/ubench/io_stress.py --reps 4000 --configs c5 --batches 655360 2>&1 | grep "^{" | sed "s/^/fixed /" | tee $out/io_stress_fixed.txt
timeout 1200 python scripts/ubench/io_stress.py --reps 1500 --configs c5r,eq_n20 --batches 393216,655360 2>&1 | grep "^{" | sed "s/^/fixed /" | tee -a $out/io_stress_fixed.txt
timeout 1200 python scripts/ubench/io_stress.py --reps 1000 --configs id_n24,id_n30_many,n32,c3 --batches 393216,655360 2>&1 | grep "^{" | sed "s/^/fixed /" | tee -a $out/io_stress_fixed.txt
