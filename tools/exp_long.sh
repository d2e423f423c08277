set -ex
mkdir -p gpurun_out/$1
python bench.py --gpus 1 --steps 20 --warmup 2 > gpurun_out/$1/bench20.json 2> gpurun_out/$1/bench20.err
python -c "
import json; d=json.load(open('gpurun_out/$1/bench20.json')); r=d['roofline']
print(d['value'], d['ms_per_step'], d['steps'], r['frac'], r['avg_launch_ms'], r['traffic'], d['parity']['ok'], d['parity']['branch_flips'], d['cpu_baseline']['value'])"
