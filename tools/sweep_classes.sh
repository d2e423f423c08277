#!/bin/bash
# Sweep of the class boundaries on config 2: bash tools/sweep_classes.sh <tag> "<split values>" "<tiny values>" [genes]
# (DN_SPLIT_LEN / DN_TINY_LEN override the capacity-derived defaults of dn_api.hip; one bench line per setting;
#  genes: total genes on the one GPU, e.g. 2500 = the shard of an 8-GPU run)
tag=$1; mkdir -p gpurun_out/$tag
genes=${4:-0}
run() {
  env "$@" python3 bench.py --genes $genes --steps 3 --warmup 1 --cpu-sample 0 --parity-genes 0 --no-also --no-end-to-end --no-rccl 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']
print('$*', round(d['value'],1), 'genes/s', round(d['ms_per_step'],1), 'ms/step  split', r['split_length'], 'pair', r['pair_length'], 'dominant', r['genes_in_kernel'], round(r['avg_launch_ms'],2), [(c['genes'], round(c['avg_launch_ms'],2)) for c in r['concurrent_kernels']])" | tee -a gpurun_out/$tag/classes.log
}
run A=1
for s in $2; do run DN_SPLIT_LEN=$s; done
for t in $3; do run DN_TINY_LEN=$t; done
