mkdir -p gpurun_out/r4q
python3 -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r4q/smoke.txt 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/r4q/smoke.txt
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q > gpurun_out/r4q/gpu_tests.txt 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r4q/gpu_tests.txt
timeout -k 10 600 python3 bench.py > gpurun_out/r4q/bench.json 2> gpurun_out/r4q/bench.err; echo "bench rc=$?"
python3 -c "
import json
d=json.load(open('gpurun_out/r4q/bench.json')); r=d['roofline']
print('c2', round(d['value'],1), round(d['ms_per_step'],1), 'frac', round(r['frac'],4), 'traffic', r['traffic'], r['traffic_info'].get('refused'), 'rccl', d['rccl']['collective_inside_library'])
c=d['also']['config 4']; r=c['roofline']; print('c4', round(c['value']), round(c['ms_per_step'],2), round(r['frac'],3), r['iteration_kernel']['latency_bound']['frac'] if r['iteration_kernel']['latency_bound'] else None)"
