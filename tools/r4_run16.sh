mkdir -p gpurun_out/r4p
timeout -k 10 300 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "randomised or class_lengths" > gpurun_out/r4p/gpu_tests.txt 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r4p/gpu_tests.txt
python3 bench.py --steps 2 --warmup 1 --cpu-sample 0 --parity-genes 0 --no-also --no-end-to-end --dump-traces gpurun_out/r4p/traces > gpurun_out/r4p/bench.json 2> gpurun_out/r4p/bench.err; echo "bench rc=$?"
python3 -c "
import json; d=json.load(open('gpurun_out/r4p/bench.json')); print(d['value'], d['roofline']['power_steps_per_solve'])"
python3 tools/redeal_study.py gpurun_out/r4p/traces.c2.npz 2 4 8 > gpurun_out/r4p/redeal_study.txt 2>&1; cat gpurun_out/r4p/redeal_study.txt
