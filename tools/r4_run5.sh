mkdir -p gpurun_out/r4f
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > gpurun_out/r4f/gpu_tests.txt 2>&1; echo "tests rc=$?"; tail -12 gpurun_out/r4f/gpu_tests.txt
bash tools/c2_ab.sh r4f tree
