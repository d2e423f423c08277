set -ex
tag=$1; shift
mkdir -p gpurun_out/$tag
for v in "$@"; do
DN_LIB_PATH=build_variants/lib_$v.so timeout -k 10 400 python tools/tier_fit.py 4000 > gpurun_out/$tag/fit_$v.log 2>&1
cat gpurun_out/$tag/fit_$v.log
done
