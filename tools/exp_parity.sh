set -ex
mkdir -p gpurun_out/$1
for g in 700 3000 9000; do
python bench.py --genes $g --steps 1 --warmup 1 --cpu-sample 0 --parity-genes 240 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('genes $g', round(d['value'],1), d['parity'])" >> gpurun_out/$1/parity.txt
done
cat gpurun_out/$1/parity.txt | cut -c1-330
