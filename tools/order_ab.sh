for s in 1700 2047 2400 2800; do
    export DN_SPLIT_LEN=$s
    python bench.py --cpu-sample 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('split $s', round(d['value'],1), 'genes/s', round(d['ms_per_step'],1), 'ms/step', round(d['roofline']['avg_launch_ms'],1), round(d['roofline']['second_kernel']['launch_to_end_ms'],1))"
done
