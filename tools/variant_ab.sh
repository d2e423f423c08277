#!/bin/bash
# A/B of prebuilt library variants (experiment helper): build each variant with `python -m degnorm_amd.build --force`,
# copy degnorm_amd/libdegnorm_amd.so to build_variants/lib_<name>.so, then: bash tools/variant_ab.sh base new base new
cp degnorm_amd/libdegnorm_amd.so /tmp/lib_keep.so
for v in "$@"; do
  cp build_variants/lib_$v.so degnorm_amd/libdegnorm_amd.so
  python bench.py --steps 1 --warmup 1 --cpu-sample 0 --parity-genes 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', round(d['value'],1), 'genes/s', round(d['ms_per_step'],1), 'ms/step', round(d['roofline']['avg_launch_ms'],1), [round(c['avg_launch_ms'],1) for c in d['roofline']['concurrent_kernels']])"
done
cp /tmp/lib_keep.so degnorm_amd/libdegnorm_amd.so
