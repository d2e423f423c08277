#!/bin/bash
# A/B of library variants on config 2: bash tools/c2_ab.sh <tag> <variant> [<variant> ...]    ('tree' = the product library;
# others: build_variants/lib_<variant>.so, see tools/build_variant.sh).  One bench line per variant -> gpurun_out/<tag>/ab.log
tag=$1; shift
mkdir -p gpurun_out/$tag
for v in "$@"; do
  if [ "$v" = "tree" ]; then unset DN_LIB_PATH; else export DN_LIB_PATH=build_variants/lib_$v.so; fi
  python3 bench.py --steps 3 --warmup 1 --cpu-sample 0 --no-also --no-end-to-end --no-rccl 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']
print('$v', round(d['value'],1), 'genes/s', round(d['ms_per_step'],1), 'ms/step  dominant', round(r['avg_launch_ms'],2), [round(c['avg_launch_ms'],2) for c in r['concurrent_kernels']], 'parity', d['parity']['ok'], d['parity']['branch_flips'], '%.1e' % d['parity']['max_rel_di'])" | tee -a gpurun_out/$tag/ab.log
done
