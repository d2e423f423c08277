set -ex
tag=$1
mkdir -p gpurun_out/$tag
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/$tag/pytest.log 2>&1 || (tail -40 gpurun_out/$tag/pytest.log; exit 1)
tail -2 gpurun_out/$tag/pytest.log
python bench.py > gpurun_out/$tag/bench.json 2> gpurun_out/$tag/bench.err
python bench.py --config c4 > gpurun_out/$tag/bench_c4.json 2> gpurun_out/$tag/bench_c4.err
python -c "
import json
for f in ('bench','bench_c4'):
    d=json.load(open('gpurun_out/$tag/%s.json' % f)); print(f, d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_ms'], d['parity']['ok'], d['parity']['branch_flips'], d['parity']['max_rel_di'])"
