set -ex
mkdir -p gpurun_out/$1
for g in 2500 5000 10000 20000; do
python bench.py --genes $g --steps 3 --warmup 1 --cpu-sample 0 --parity-genes 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('genes per GPU $g: %.1f genes/s %.1f ms per run %.1f ms per sweep' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms']))" >> gpurun_out/$1/shard.txt
done
cat gpurun_out/$1/shard.txt
