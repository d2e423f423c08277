mkdir -p gpurun_out/r4e
bash tools/c2_ab.sh r4e tree mfma tree
timeout -k 10 400 python3 -m pytest tests -m gpu -x -q > gpurun_out/r4e/gpu_tests.txt 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/r4e/gpu_tests.txt
DN_LIB_PATH=build_variants/lib_st_mfma.so python3 tools/trace_stats.py 4000 > gpurun_out/r4e/phase_mfma.txt 2>&1
DN_LIB_PATH=build_variants/lib_st_dpp.so python3 tools/trace_stats.py 4000 > gpurun_out/r4e/phase_dpp.txt 2>&1
grep "per inner iteration" gpurun_out/r4e/phase_mfma.txt gpurun_out/r4e/phase_dpp.txt
