mkdir -p gpurun_out/r4y
timeout -k 10 300 python3 tools/init_ab.py 16000 tree > gpurun_out/r4y/init_ab.txt 2>&1; cat gpurun_out/r4y/init_ab.txt
DN_LIB_PATH=build_variants/lib_stamp.so timeout -k 10 300 python3 tools/init_phases.py 16000 > gpurun_out/r4y/init_phases.txt 2>&1; tail -7 gpurun_out/r4y/init_phases.txt
timeout -k 10 700 python3 -m pytest tests -m gpu -x -q > gpurun_out/r4y/gpu_tests.txt 2>&1 || { echo "tests failed"; tail -30 gpurun_out/r4y/gpu_tests.txt; exit 1; }
tail -3 gpurun_out/r4y/gpu_tests.txt
python3 bench.py --config c4 --cpu-sample 0 > gpurun_out/r4y/c4_bench.json 2> gpurun_out/r4y/c4.err && \
python3 -c "
import json
c=json.load(open('gpurun_out/r4y/c4_bench.json')); r=c['roofline']
print('c4', round(c['value']), round(c['ms_per_step'],2), 'init', round(r['avg_launch_ms'],2), r['frac'], 'iter', r['iteration_kernel']['avg_launch_ms'], c['parity']['ok'])"
