# round 4, final: GPU suite, bench lines, profiles (forward + backward) of the round's last binary
scripts/gpu_round.sh r04 tests
scripts/gpu_round.sh r04 bench
PROF_CONFIGS="c3 c5" scripts/gpu_round.sh r04 prof
scripts/gpu_round.sh r04 bwd
timeout 300 python scripts/ubench/small_batch.py > gpurun_out/r04/small_batch.txt 2>&1
