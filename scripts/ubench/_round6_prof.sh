# profile one bench configuration under rocprofv3, keep the summary only:  _round6_prof.sh <tag> <out.json> <note> [bench args]
tag=$1; outjson=$2; note=$3; shift 3
bash scripts/profile_bench.sh $tag "$@" > /dev/null 2>&1
mkdir -p gpurun_out/r06prof
python scripts/summarize_profile.py $tag gpurun_out/r06prof/$outjson "$note" > gpurun_out/r06prof/$tag.log 2>&1
tail -2 gpurun_out/r06prof/$tag.log
rm -rf gpurun_out/prof_$tag
