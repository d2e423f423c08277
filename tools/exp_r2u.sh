set -ex
mkdir -p gpurun_out/r2u
DN_LIB_PATH=build_variants/lib_stamp.so timeout -k 10 300 python tools/trace_stats.py 4000 > gpurun_out/r2u/ts.log 2>&1
tail -8 gpurun_out/r2u/ts.log
