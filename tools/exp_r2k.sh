set -ex
mkdir -p gpurun_out/r2k
for v in base noload nowrite nogram; do
  DN_LIB_PATH=build_variants/lib_x_$v.so timeout -k 10 200 python tools/trace_stats.py 256 100 1400 > gpurun_out/r2k/ts_$v.log 2>&1
  echo "== $v"; grep -E "per inner iteration:|pass cycles" gpurun_out/r2k/ts_$v.log
done
