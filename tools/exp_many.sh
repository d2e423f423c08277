# usage: bash tools/exp_many.sh <tag> <variant> <steps>   -- one long bench run; prints per-class averages and the worst sweep span
set -ex
mkdir -p gpurun_out/$1
DN_LIB_PATH=build_variants/lib_$2.so python bench.py --steps $3 --warmup 1 --cpu-sample 0 --parity-genes 0 > gpurun_out/$1/b_$2.json 2>/dev/null
python -c "
import json; d=json.load(open('gpurun_out/$1/b_$2.json')); r=d['roofline']
print('$2', d['value'], d['ms_per_step'], r['avg_launch_ms'], [(c['kernel'], round(c['avg_launch_ms'],2)) for c in r['concurrent_kernels']])"
