# usage: bash tools/exp_py.sh <tag> <script.py> [args]   -- run a diagnostic script on the GPU box, output to gpurun_out/<tag>/
set -ex
tag=$1; shift
mkdir -p gpurun_out/$tag
timeout -k 10 500 python "$@" > gpurun_out/$tag/out.log 2>&1 || (tail -20 gpurun_out/$tag/out.log; exit 1)
tail -30 gpurun_out/$tag/out.log
