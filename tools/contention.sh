set -e
cp degnorm_amd/libdegnorm_amd.so /tmp/lib_keep.so
for v in nt2 nt1; do
  cp build_variants/lib_$v.so degnorm_amd/libdegnorm_amd.so
  echo "variant $v: genes 1024 L 5000"; timeout -k 10 200 python tools/trace_stats.py 1024 100 5000 2>&1 | grep -E "launch ms|per inner iteration:|pass cycles" | tail -3
  echo "variant $v mixed"; timeout -k 10 200 python tools/trace_stats.py 4000 100 2>&1 | grep -E "launch ms" | tail -1
done
cp /tmp/lib_keep.so degnorm_amd/libdegnorm_amd.so
