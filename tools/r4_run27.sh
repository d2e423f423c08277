mkdir -p gpurun_out/r4B
timeout -k 10 700 python3 -m pytest tests -m gpu -x -q > gpurun_out/r4B/gpu_tests.txt 2>&1 || { echo "tests failed"; tail -30 gpurun_out/r4B/gpu_tests.txt; exit 1; }
tail -3 gpurun_out/r4B/gpu_tests.txt
python3 bench.py --no-also --cpu-sample 0 > gpurun_out/r4B/c2_bench.json 2> gpurun_out/r4B/c2.err && \
python3 -c "
import json
c=json.load(open('gpurun_out/r4B/c2_bench.json')); r=c['roofline']
print('c2', round(c['value']), round(c['ms_per_step'],2), 'sweep', r.get('avg_launch_ms'), r['frac'], c['parity'].get('ok'))"
