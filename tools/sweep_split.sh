#!/bin/bash
# Sweep of the narrow / wide class boundary (DN_SPLIT_LEN overrides the 2.1 x LDS-columns default of dn_api.hip).
for s in "$@"; do
  DN_SPLIT_LEN=$s python bench.py --steps 1 --warmup 1 --cpu-sample 0 --parity-genes 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('split $s', round(d['value'],1), 'genes/s', round(d['ms_per_step'],1), 'ms/step', round(d['roofline']['avg_launch_ms'],1), [round(c['avg_launch_ms'],1) for c in d['roofline']['concurrent_kernels']])"
done
