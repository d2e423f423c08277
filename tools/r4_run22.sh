mkdir -p gpurun_out/r4v
timeout -k 10 700 python3 -m pytest tests -m gpu -x -q > gpurun_out/r4v/gpu_tests.txt 2>&1 || { echo "tests failed"; tail -30 gpurun_out/r4v/gpu_tests.txt; exit 1; }
tail -3 gpurun_out/r4v/gpu_tests.txt
python3 bench.py --config c4 --cpu-sample 0 > gpurun_out/r4v/c4_bench.json 2> gpurun_out/r4v/c4.err && \
python3 bench.py --no-also --cpu-sample 0 > gpurun_out/r4v/c2_bench.json 2> gpurun_out/r4v/c2.err && \
python3 -c "
import json
c=json.load(open('gpurun_out/r4v/c4_bench.json')); r=c['roofline']
print('c4', round(c['value']), round(c['ms_per_step'],2), 'init', round(r['avg_launch_ms'],2), 'iter', r['iteration_kernel']['avg_launch_ms'], c['parity']['ok'])
c=json.load(open('gpurun_out/r4v/c2_bench.json')); r=c['roofline']
print('c2', round(c['value']), round(c['ms_per_step'],2), 'sweep', r.get('avg_launch_ms'), r['frac'], c['parity'].get('ok'))"
