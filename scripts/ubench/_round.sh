# round 4, closing: the failing shape of the repetition fault, 30 000 launches (15 000 with the record) on the library in the tree
out=gpurun_out/r04z; mkdir -p $out
timeout 1500 python scripts/ubench/io_stress.py --reps 30000 --configs c5 --batches 655360 2>&1 | grep "^{" | tee $out/io_stress_30000.txt
timeout 900 python scripts/ubench/io_stress.py --reps 10000 --configs c5r --batches 655360 2>&1 | grep "^{" | tee -a $out/io_stress_30000.txt
