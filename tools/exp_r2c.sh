set -x
mkdir -p gpurun_out/r2c
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/r2c/pytest.log 2>&1
python bench.py --steps 2 > gpurun_out/r2c/bench_c2.json 2> gpurun_out/r2c/bench_c2.err
python bench.py --config c4 --steps 2 > gpurun_out/r2c/bench_c4.json 2> gpurun_out/r2c/bench_c4.err
DN_LIB_PATH=build_variants/lib_stamp.so timeout -k 10 300 python tools/trace_stats.py 4000 > gpurun_out/r2c/ts_256.log 2>&1
DN_LIB_PATH=build_variants/lib_w512.so timeout -k 10 300 python tools/trace_stats.py 4000 > gpurun_out/r2c/ts_512.log 2>&1
DN_LIB_PATH=build_variants/lib_stamp.so python bench.py --cpu-sample 0 --parity-genes 0 > gpurun_out/r2c/bench_stamp256.json 2>&1
DN_LIB_PATH=build_variants/lib_w512.so python bench.py --cpu-sample 0 --parity-genes 0 > gpurun_out/r2c/bench_stamp512.json 2>&1
tail -3 gpurun_out/r2c/pytest.log
