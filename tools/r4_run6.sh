mkdir -p gpurun_out/r4g
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q > gpurun_out/r4g/gpu_tests.txt 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r4g/gpu_tests.txt
timeout -k 10 600 python3 bench.py > gpurun_out/r4g/bench.json 2> gpurun_out/r4g/bench.err; echo "bench rc=$?"; tail -3 gpurun_out/r4g/bench.err
python3 - <<'PY'
import json
d=json.load(open('gpurun_out/r4g/bench.json'))
r=d['roofline']
print('c2', d['value'], d['ms_per_step'], 'frac', r['frac'], 'sweep', r['sweep_ms'], 'stream', r['stream_read_ceiling_gbps'], d['rccl'])
print('parity', {k:(v if not isinstance(v,dict) else {kk:vv for kk,vv in v.items() if kk in ('flipped_genes','branch_flips','genes','tie_sensitive_genes','max_rel_di','max_rel_di_final','ok')}) for k,v in d['parity'].items() if k!='what'})
c=d['also']['config 4']; r=c['roofline']
print('c4', c['value'], c['ms_per_step'], 'init frac', r['frac'], r['avg_launch_ms'], r['frac_of_stream_read_ceiling'], r['row_maxima_kernel'], r['iteration_kernel']['avg_launch_ms'])
print('e2e', d['end_to_end']['genes_per_s'], 'cpu', d['cpu_baseline']['value'], d['cpu_baseline']['cores'])
PY
