#!/bin/bash
# One gpurun call of the round: GPU test suite, per-config bench lines, rocprofv3 stats + PMC per config.
#   scripts/gpu_round.sh <tag> [tests|bench|prof|bwd|mapper ...]
tag=${1:-r03}; shift || true
what=${*:-tests bench prof}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
for w in $what; do
case $w in
tests)
  timeout 2400 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider > $out/pytest.log 2>&1
  echo "pytest rc=$?" >> $out/pytest.log
  tail -5 $out/pytest.log ;;
bench)
  for cfg in c3 c1 c2 c4 c5 c5r; do
    timeout 600 python bench.py --config $cfg > $out/bench_${cfg}_fp32.json 2> $out/bench_${cfg}_fp32.err
  done
  timeout 600 python bench.py --config c3 --dtype fp64 > $out/bench_c3_fp64.json 2> $out/bench_c3_fp64.err
  timeout 600 python bench.py --config c4 --dtype fp64 > $out/bench_c4_fp64.json 2> $out/bench_c4_fp64.err
  timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $out/bench_c3_driver_like.json 2>&1
  timeout 600 python bench.py --force-dist --no-cpu-baseline > $out/bench_c3_forcedist.json 2>&1
  head -c 600 $out/bench_c3_fp32.json; echo ;;
prof)
  for cfg in ${PROF_CONFIGS:-c3 c4 c2 c5 c1}; do
    timeout 900 scripts/profile_bench.sh ${tag}_$cfg --config $cfg > $out/prof_$cfg.log 2>&1
    timeout 120 python scripts/summarize_profile.py ${tag}_$cfg $out/prof_${cfg}_summary.json "config $cfg, fp32, default kernels" >> $out/prof_$cfg.log 2>&1
    rm -rf gpurun_out/prof_${tag}_$cfg          # raw rocprofv3 databases: ~20 MB per config, gpurun_out/ is capped at 64 MiB
  done ;;
bwd)
  timeout 600 python scripts/ubench/bwd_c5.py c5 c5r > $out/bwd_c5.txt 2> $out/bwd_c5.err
  timeout 600 python scripts/ubench/bwd_split.py > $out/backward_split.txt 2>&1
  for cfg in c5 c5r c3; do
    timeout 600 python scripts/ubench/bwd_profile_summary.py $cfg $out/prof_backward_$cfg.json > /dev/null 2>&1
  done
  timeout 2000 python scripts/ubench/lmi_sweep.py --oracle 2>&1 | grep "^{" > $out/lmi_sweep.txt
  cat $out/bwd_c5.txt ;;
mapper)
  timeout 600 python bench.py --mapper 64 --no-cpu-baseline > $out/bench_c3_mapper64_fused.json 2> $out/bench_c3_mapper64_fused.err
  timeout 600 python bench.py --mapper 64 --no-fuse --no-cpu-baseline > $out/bench_c3_mapper64_twoop.json 2> $out/bench_c3_mapper64_twoop.err
  timeout 600 python bench.py --config c5 --mapper 32 --no-cpu-baseline > $out/bench_c5_mapper32_fused.json 2>&1
  timeout 600 python bench.py --config c5 --mapper 32 --no-fuse --no-cpu-baseline > $out/bench_c5_mapper32_twoop.json 2>&1
  cat $out/bench_c3_mapper64_fused.json | head -c 400; echo ;;
esac
done
