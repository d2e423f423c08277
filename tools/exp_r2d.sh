set -ex
mkdir -p gpurun_out/r2d
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/r2d/pytest.log 2>&1
tail -3 gpurun_out/r2d/pytest.log
python bench.py --steps 2 --cpu-sample 0 > gpurun_out/r2d/bench_c2.json 2> gpurun_out/r2d/bench_c2.err
DN_LIB_PATH=build_variants/lib_stamp.so timeout -k 10 300 python tools/trace_stats.py 4000 > gpurun_out/r2d/ts_rt.log 2>&1
tail -14 gpurun_out/r2d/ts_rt.log
