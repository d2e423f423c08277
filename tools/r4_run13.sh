mkdir -p gpurun_out/r4m
run() {
  env "$@" python3 bench.py --genes $G --steps 3 --warmup 1 --cpu-sample 0 --parity-genes 0 --no-also --no-end-to-end --no-rccl 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']
print('genes $G', '$*', round(d['value'],1), 'genes/s', round(d['ms_per_step'],1), 'ms/step  split', r['split_length'], 'pair', r['pair_length'], round(r['avg_launch_ms'],2), [(c['genes'], round(c['avg_launch_ms'],2)) for c in r['concurrent_kernels']])" | tee -a gpurun_out/r4m/classes.log
}
G=0
run DN_SPLIT_LEN=3700 DN_TINY_LEN=1700
run DN_SPLIT_LEN=3600 DN_TINY_LEN=1700
run DN_SPLIT_LEN=3700 DN_TINY_LEN=1600
run DN_SPLIT_LEN=3800 DN_TINY_LEN=1750
G=2500
run A=1
run DN_SPLIT_LEN=3700 DN_TINY_LEN=1700
run DN_SPLIT_LEN=3500 DN_TINY_LEN=1700
G=10000
run A=1
run DN_SPLIT_LEN=3700 DN_TINY_LEN=1700
