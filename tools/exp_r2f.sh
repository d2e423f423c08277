set -ex
mkdir -p gpurun_out/r2f
DN_LIB_PATH=build_variants/lib_stamp.so timeout -k 10 300 python tools/trace_stats.py 4000 > gpurun_out/r2f/ts.log 2>&1
tail -16 gpurun_out/r2f/ts.log
bash tools/profile_round.sh r2f_prof
cat gpurun_out/r2f_prof/bench.json | cut -c1-400
