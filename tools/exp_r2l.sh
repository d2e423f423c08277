set -ex
mkdir -p gpurun_out/r2l build_variants
cp degnorm_amd/libdegnorm_amd.so build_variants/lib_pf2.so
bash tools/variant_ab.sh noraw pf2 noraw pf2 > gpurun_out/r2l/ab.log 2>&1
cat gpurun_out/r2l/ab.log
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "golden or edge" > gpurun_out/r2l/pytest.log 2>&1 || (tail -30 gpurun_out/r2l/pytest.log; exit 1)
tail -2 gpurun_out/r2l/pytest.log
