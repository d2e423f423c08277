mkdir -p gpurun_out/r4D
timeout -k 10 700 python3 -m pytest tests -m gpu -x -q > gpurun_out/r4D/gpu_tests.txt 2>&1 || { echo "tests failed"; tail -30 gpurun_out/r4D/gpu_tests.txt; exit 1; }
tail -3 gpurun_out/r4D/gpu_tests.txt
timeout -k 10 300 python3 tools/rows_latency.py gpurun_out/r4D/rows_inner_iteration_cycles.json > gpurun_out/r4D/rows_latency.txt 2>&1; tail -3 gpurun_out/r4D/rows_latency.txt
python3 bench.py > gpurun_out/r4D/bench_default.json 2> gpurun_out/r4D/bench_default.err && \
python3 -c "
import json
c=json.load(open('gpurun_out/r4D/bench_default.json')); r=c['roofline']
print('c2', round(c['value']), round(c['ms_per_step'],2), 'sweep', r.get('avg_launch_ms'), r['frac'], 'traffic', r.get('traffic'), c['parity'].get('ok'))
a=c.get('also',{})
for k,v in a.items(): print(k, {kk: v[kk] for kk in v if kk in ('value','ms_per_step')}, v.get('roofline',{}).get('frac'), v.get('roofline',{}).get('traffic'))
print('cpu_baseline', c.get('cpu_baseline'))"
