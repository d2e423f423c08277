set -ex
mkdir -p gpurun_out/r2m
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r2m/pytest.log 2>&1 || (tail -40 gpurun_out/r2m/pytest.log; exit 1)
tail -2 gpurun_out/r2m/pytest.log
bash tools/variant_ab.sh noraw x16 noraw x16 > gpurun_out/r2m/ab.log 2>&1
cat gpurun_out/r2m/ab.log
