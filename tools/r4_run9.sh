mkdir -p gpurun_out/r4j
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q > gpurun_out/r4j/gpu_tests.txt 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r4j/gpu_tests.txt
timeout -k 10 600 python3 bench.py --gpus 2 --backend gloo --sharded-api --redeal --steps 2 --warmup 1 > gpurun_out/r4j/rehearsal_c2_2ranks.json 2> gpurun_out/r4j/rehearsal.err; echo "rehearsal rc=$?"; tail -2 gpurun_out/r4j/rehearsal.err
python3 - <<'PY'
import json
d=json.load(open('gpurun_out/r4j/rehearsal_c2_2ranks.json'))
print(d['value'], d['ms_per_step'], d['n_gpus'], d['config']['redeal'], d['parity']['ok'] if d.get('parity') else None)
e=d['end_to_end_sharded']
print({k:v for k,v in e.items() if k not in ('what','per_rank')})
for r in e['per_rank']: print(r)
PY
timeout -k 10 600 python3 bench.py --gpus 4 --backend gloo --redeal --steps 2 --warmup 1 --config c4 > gpurun_out/r4j/rehearsal_c4_4ranks.json 2> gpurun_out/r4j/rehearsal4.err; echo "rehearsal c4 rc=$?"
python3 -c "
import json
d=json.load(open('gpurun_out/r4j/rehearsal_c4_4ranks.json')); print(d['value'], d['ms_per_step'], d['config']['redeal'], d['parity']['ok'] if d.get('parity') else None)
from degnorm_amd import _lib
dv=_lib.Device(0); print('read ceiling', dv.measure_read_gbps(1<<30,5))"
