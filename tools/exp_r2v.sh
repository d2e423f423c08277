set -ex
mkdir -p gpurun_out/r2v
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r2v/pytest.log 2>&1 || (tail -40 gpurun_out/r2v/pytest.log; exit 1)
tail -2 gpurun_out/r2v/pytest.log
bash tools/variant_ab.sh x16 bk x16 bk > gpurun_out/r2v/ab.log 2>&1
cat gpurun_out/r2v/ab.log
