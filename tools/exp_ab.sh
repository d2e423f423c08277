# usage: bash tools/exp_ab.sh <tag> <variant> <variant> ...   (A/B of prebuilt build_variants/lib_<variant>.so)
set -ex
tag=$1; shift
mkdir -p gpurun_out/$tag
bash tools/variant_ab.sh "$@" > gpurun_out/$tag/ab.log 2>&1
cat gpurun_out/$tag/ab.log
