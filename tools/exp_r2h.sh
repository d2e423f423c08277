set -ex
mkdir -p gpurun_out/r2h
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r2h/pytest.log 2>&1 || (tail -40 gpurun_out/r2h/pytest.log; exit 1)
tail -3 gpurun_out/r2h/pytest.log
python bench.py --steps 3 --cpu-sample 0 > gpurun_out/r2h/bench_c2.json 2> gpurun_out/r2h/bench_c2.err
python bench.py --config c4 --steps 3 --cpu-sample 0 > gpurun_out/r2h/bench_c4.json 2> gpurun_out/r2h/bench_c4.err
python - <<'PY'
import json
for f in ('gpurun_out/r2h/bench_c2.json', 'gpurun_out/r2h/bench_c4.json'):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(f, round(d['value'], 1), round(d['ms_per_step'], 1), d['roofline']['kernel'], round(d['roofline']['avg_launch_ms'], 2), d['parity']['ok'], d['parity']['max_rel_di'], d['parity']['branch_flips'])
PY
