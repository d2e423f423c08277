# usage: bash tools/exp_env.sh <tag> <genes> "ENV=val ..." "ENV=val ..." ...   -- bench line under several environments
set -e
tag=$1; genes=$2; shift; shift
mkdir -p gpurun_out/$tag
for e in "$@"; do
for rep in 1 2; do
env $e python bench.py --genes $genes --steps 3 --warmup 1 --cpu-sample 0 --parity-genes 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('[$e] %.1f genes/s %.1f ms per run %.1f ms per sweep' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms']), [round(c['avg_launch_ms'],1) for c in d['roofline']['concurrent_kernels']])" >> gpurun_out/$tag/env.log
done
done
cat gpurun_out/$tag/env.log
