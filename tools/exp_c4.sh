# usage: bash tools/exp_c4.sh <tag> <variant> ...   -- config 4 bench line (p = 50, take-every 500) per library variant
set -ex
tag=$1; shift
mkdir -p gpurun_out/$tag
for v in "$@"; do
DN_LIB_PATH=build_variants/lib_$v.so python bench.py --config c4 --steps 2 --warmup 1 --cpu-sample 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']
print('$v', round(d['value'],1), 'genes/s', round(d['ms_per_step'],1), 'ms/step  sweep', round(r['iteration_kernel']['avg_launch_ms'],2), 'init', round(r['avg_launch_ms'],2), 'parity', d['parity']['ok'] if d.get('parity') else None, d['parity']['max_rel_di'] if d.get('parity') else None)" >> gpurun_out/$tag/c4.log
done
cat gpurun_out/$tag/c4.log
