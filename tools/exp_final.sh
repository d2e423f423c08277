# the round's closing run on the GPU box: default bench line (-> profiles/<round>/c2_bench.json) and the entry-point smoke test
set -ex
mkdir -p gpurun_out/final
python bench.py > gpurun_out/final/c2_bench.json 2> gpurun_out/final/c2_bench.err
python -c "
import json
d=json.load(open('gpurun_out/final/c2_bench.json')); r=d['roofline']
print('value', d['value'], 'ms/step', d['ms_per_step'], 'frac', r['frac'], 'sweep', r['avg_launch_ms'], 'traffic', r['traffic'], r['traffic_info'].get('refused'), 'parity', d['parity']['ok'], d['parity']['branch_flips'])"
python -c "import __graft_entry__ as g; g.smoke()"
