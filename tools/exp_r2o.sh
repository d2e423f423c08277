set -ex
mkdir -p gpurun_out/r2o
DN_LIB_PATH=build_variants/lib_stamp.so timeout -k 10 300 python tools/trace_stats.py 4000 > gpurun_out/r2o/ts.log 2>&1
tail -17 gpurun_out/r2o/ts.log
for L in 1200 2200 3000 4400; do echo "== L $L"; DN_LIB_PATH=build_variants/lib_stamp.so timeout -k 10 200 python tools/trace_stats.py 512 100 $L 2>&1 | grep -E "launch ms|per inner iteration:|pass cycles" | tail -3; done > gpurun_out/r2o/fixed.log 2>&1
cat gpurun_out/r2o/fixed.log
