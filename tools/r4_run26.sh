mkdir -p gpurun_out/r4z
timeout -k 10 700 python3 -m pytest tests -m gpu -x -q > gpurun_out/r4z/gpu_tests_full.txt 2>&1 || { echo "tests failed"; tail -30 gpurun_out/r4z/gpu_tests_full.txt; exit 1; }
tail -3 gpurun_out/r4z/gpu_tests_full.txt
python3 bench.py --config c4 --cpu-sample 0 > gpurun_out/r4z/c4_bench.json 2> gpurun_out/r4z/c4.err && \
python3 -c "
import json
c=json.load(open('gpurun_out/r4z/c4_bench.json')); r=c['roofline']
print('c4', round(c['value']), round(c['ms_per_step'],2), 'init', round(r['avg_launch_ms'],2), r['frac'], 'iter', r['iteration_kernel']['avg_launch_ms'], c['parity']['ok'])"
