# usage: bash tools/exp_full.sh <tag>   -- GPU test suite, then one default bench line, on the in-tree library
set -ex
tag=$1
mkdir -p gpurun_out/$tag
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/$tag/pytest.log 2>&1 || (tail -40 gpurun_out/$tag/pytest.log; exit 1)
tail -2 gpurun_out/$tag/pytest.log
python bench.py > gpurun_out/$tag/bench.json 2> gpurun_out/$tag/bench.err
python -c "
import json; d=json.load(open('gpurun_out/$tag/bench.json'))
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_ms'], d['parity'], d['cpu_baseline']['value'])"
