set -ex
tag=$1; v=$2; shift; shift
mkdir -p gpurun_out/$tag
DN_LIB_PATH=build_variants/lib_$v.so timeout -k 10 400 python tools/trace_stats.py "$@" > gpurun_out/$tag/ts_$v.log 2>&1
tail -12 gpurun_out/$tag/ts_$v.log
