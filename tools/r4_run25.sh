mkdir -p gpurun_out/r4A
timeout -k 10 300 python3 tools/init_ab.py 16000 tree > gpurun_out/r4A/init_ab.txt 2>&1; cat gpurun_out/r4A/init_ab.txt
DN_LIB_PATH=build_variants/lib_stamp.so timeout -k 10 300 python3 tools/init_phases.py 16000 > gpurun_out/r4A/init_phases.txt 2>&1; tail -7 gpurun_out/r4A/init_phases.txt
timeout -k 10 700 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q  > gpurun_out/r4A/gpu_tests.txt 2>&1 || { echo "tests failed"; tail -30 gpurun_out/r4A/gpu_tests.txt; exit 1; }
tail -3 gpurun_out/r4A/gpu_tests.txt
