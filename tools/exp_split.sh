set -ex
tag=$1; shift
mkdir -p gpurun_out/$tag
bash tools/sweep_split.sh "$@" > gpurun_out/$tag/split.log 2>&1
cat gpurun_out/$tag/split.log
