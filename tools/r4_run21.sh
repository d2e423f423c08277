mkdir -p gpurun_out/r4u
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q > gpurun_out/r4u/gpu_tests.txt 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r4u/gpu_tests.txt
python3 bench.py --config c4 --cpu-sample 0 > gpurun_out/r4u/c4_bench.json 2> gpurun_out/r4u/c4.err; echo "c4 rc=$?"
python3 -c "
import json
c=json.load(open('gpurun_out/r4u/c4_bench.json')); r=c['roofline']
print('c4', round(c['value']), round(c['ms_per_step'],2), 'init', round(r['avg_launch_ms'],2), 'iter', r['iteration_kernel']['avg_launch_ms'], c['parity']['ok'])"
