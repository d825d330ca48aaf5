mkdir -p gpurun_out/r06prof
for cfg in c4 c5 c2 c1; do
  bash scripts/ubench/_round6_prof.sh r06_$cfg r06_${cfg}_fp32_rocprofv3.json "config $cfg, fp32, default kernels" --config $cfg
done
for cfg in c3 c5; do timeout 600 python scripts/ubench/bwd_profile_summary.py $cfg gpurun_out/r06prof/r06_${cfg}_backward_rocprofv3.json; done
ls gpurun_out/r06prof
