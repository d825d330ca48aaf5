#!/bin/bash
# rocprofv3 kernel-trace statistics of scripts/kernel_suite.py -> gpurun_out/prof_<tag>/suite
set -u
tag=${1:-suite}
out=$PWD/gpurun_out/prof_$tag
mkdir -p "$out"
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d "$out/suite" -o stats -- python scripts/kernel_suite.py > "$out/suite.log" 2> "$out/suite.err"
ls "$out/suite"
