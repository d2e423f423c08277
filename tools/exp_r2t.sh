set -ex
mkdir -p gpurun_out/r2t
for g in 2500 5000 10000; do python bench.py --genes $g --steps 3 --cpu-sample 0 --parity-genes 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('genes per GPU $g:', round(d['value'],1), 'genes/s', round(d['ms_per_step'],1), 'ms per run', round(d['roofline']['avg_launch_ms'],1), 'ms per sweep')"; done > gpurun_out/r2t/shard_sizes.log 2>&1
cat gpurun_out/r2t/shard_sizes.log
