mkdir -p gpurun_out/r4t
timeout -k 10 600 python3 bench.py --gpus 2 --backend gloo --sharded-api --redeal --steps 2 --warmup 1 > gpurun_out/r4t/rehearsal_c2_2ranks.json 2> gpurun_out/r4t/rehearsal.err; echo "rehearsal rc=$?"
python3 - <<'PY'
import json
d=json.load(open('gpurun_out/r4t/rehearsal_c2_2ranks.json'))
print(d['value'], d['ms_per_step'], d['config']['redeal'], d['parity']['ok'])
e=d['end_to_end_sharded']
print({k:v for k,v in e.items() if k not in ('what','per_rank')})
for r in e['per_rank']: print(r)
PY
