set -ex
mkdir -p gpurun_out/final
python bench.py > gpurun_out/final/c2_bench.json 2> gpurun_out/final/c2_bench.err
python -c "
import json
d=json.load(open('gpurun_out/final/c2_bench.json')); r=d['roofline']
print('value', d['value'], 'frac', r['frac'], 'traffic', r['traffic'], r['traffic_info'].get('refused'), 'parity', d['parity']['ok'])"
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/final/pytest.log 2>&1 || (tail -30 gpurun_out/final/pytest.log; exit 1)
tail -2 gpurun_out/final/pytest.log
python -c "import __graft_entry__ as g; g.smoke()"
