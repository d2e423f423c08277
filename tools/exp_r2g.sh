set -ex
mkdir -p gpurun_out/r2g
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "wide_cohorts or config4 or edge_shapes or downsampled" > gpurun_out/r2g/pytest.log 2>&1 || (tail -30 gpurun_out/r2g/pytest.log; exit 1)
tail -3 gpurun_out/r2g/pytest.log
python bench.py --config c4 --steps 2 > gpurun_out/r2g/bench_c4.json 2> gpurun_out/r2g/bench_c4.err
cut -c1-1500 gpurun_out/r2g/bench_c4.json
