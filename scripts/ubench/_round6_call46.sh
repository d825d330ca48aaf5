cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06prof
bash scripts/ubench/_round6_prof.sh r06_c3 r06_c3_fp32_rocprofv3.json "config c3, fp32, default kernels (W-in-LDS schedule, rows through buffer descriptors), buffers rotated past the Infinity Cache"
python scripts/ubench/bwd_profile_summary.py c3 gpurun_out/r06prof/r06_c3_backward_rocprofv3.json
