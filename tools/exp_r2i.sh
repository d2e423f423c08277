set -ex
mkdir -p gpurun_out/r2i build_variants
cp degnorm_amd/libdegnorm_amd.so build_variants/lib_raw.so
bash tools/variant_ab.sh noraw raw noraw raw > gpurun_out/r2i/ab.log 2>&1
cat gpurun_out/r2i/ab.log
