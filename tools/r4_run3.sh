mkdir -p gpurun_out/r4d
python3 tools/ubench/solver_ab.py 10 > gpurun_out/r4d/solver_ab.txt 2>&1; echo "ubench rc=$?"; tail -3 gpurun_out/r4d/solver_ab.txt
bash tools/c2_ab.sh r4d tree mfma tree
timeout -k 10 400 python3 -m pytest tests -m gpu -x -q > gpurun_out/r4d/gpu_tests.txt 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/r4d/gpu_tests.txt
