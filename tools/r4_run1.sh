mkdir -p gpurun_out/r4a
python3 tools/ubench/solver_ab.py 14 > gpurun_out/r4a/solver_ab.txt 2>&1; echo "ubench rc=$?"; tail -4 gpurun_out/r4a/solver_ab.txt
timeout -k 10 400 python3 -m pytest tests -m gpu -x -q > gpurun_out/r4a/gpu_tests.txt 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/r4a/gpu_tests.txt
bash tools/c2_ab.sh r4a tree mfma tree mfma
DN_LIB_PATH=build_variants/lib_raw12.so timeout -k 10 300 python3 -m pytest tests/test_gpu_parity.py -m gpu -q -k "edge_shapes or pair_class" > gpurun_out/r4a/raw12_tests.txt 2>&1; echo "raw12 rc=$?"; tail -15 gpurun_out/r4a/raw12_tests.txt
