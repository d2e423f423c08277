# usage: bash tools/exp_cls.sh <tag> <variant> <tiny_len values...>   -- class times for several pair-class boundaries
set -ex
tag=$1; v=$2; shift; shift
mkdir -p gpurun_out/$tag
for m in "$@"; do
DN_TINY_LEN=$m DN_LIB_PATH=build_variants/lib_$v.so timeout -k 10 300 python tools/class_times.py > gpurun_out/$tag/cls_${v}_$m.log 2>&1 || (tail -5 gpurun_out/$tag/cls_${v}_$m.log; exit 1)
echo "tiny_len $m"; tail -2 gpurun_out/$tag/cls_${v}_$m.log
done
