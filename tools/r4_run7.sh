mkdir -p gpurun_out/r4h
timeout -k 10 400 python3 -m pytest tests/test_gpu_sharded.py tests/test_warm_start.py tests/test_coverage_merge.py -m gpu -x -q > gpurun_out/r4h/gpu_tests.txt 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r4h/gpu_tests.txt
timeout -k 10 600 python3 bench.py --gpus 2 --backend gloo --sharded-api --steps 2 --warmup 1 > gpurun_out/r4h/rehearsal_c2_2ranks.json 2> gpurun_out/r4h/rehearsal.err; echo "rehearsal rc=$?"; tail -2 gpurun_out/r4h/rehearsal.err
python3 - <<'PY'
import json
d=json.load(open('gpurun_out/r4h/rehearsal_c2_2ranks.json'))
print(d['value'], d['ms_per_step'], d['n_gpus'])
e=d['end_to_end_sharded']
print({k:v for k,v in e.items() if k not in ('what','per_rank')})
for r in e['per_rank']: print(r)
PY
python3 -c "
from degnorm_amd import _lib
d=_lib.Device(0); print('read ceiling', d.measure_read_gbps(1<<30,5), 'copy', d.measure_copy_gbps(1<<30,5))"
