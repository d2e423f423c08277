set -ex
mkdir -p gpurun_out/r2r
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r2r/pytest.log 2>&1 || (tail -40 gpurun_out/r2r/pytest.log; exit 1)
tail -2 gpurun_out/r2r/pytest.log
python bench.py --steps 3 > gpurun_out/r2r/bench_c2.json 2> gpurun_out/r2r/bench_c2.err
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r2r/bench_c2.json').read().strip().splitlines()[-1]); r = d['roofline']
print(round(d['value'], 1), round(d['ms_per_step'], 1), r['kernel'], round(r['avg_launch_ms'], 2), r['concurrent_kernel']['avg_launch_ms'], 'frac', round(r['frac'], 3), d['parity']['ok'], d['parity']['max_rel_di'], d['cpu_baseline']['value'])
PY
