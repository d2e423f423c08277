mkdir -p gpurun_out/r4i
python3 tools/determinism_probe.py 768 tree maxilp o2 > gpurun_out/r4i/determinism.txt 2>&1; echo "det rc=$?"; cat gpurun_out/r4i/determinism.txt | tail -8
