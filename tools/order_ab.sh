for g in 20000; do
  for m in work static work static; do
    if [ $m = static ]; then export DN_STATIC_ORDER=1; else unset DN_STATIC_ORDER; fi
    python bench.py --genes $g --cpu-sample 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$g $m', round(d['value'],1), 'genes/s', round(d['ms_per_step'],1), 'ms/step', round(d['roofline']['avg_launch_ms'],1), round(d['roofline']['second_kernel']['launch_to_end_ms'],1))"
  done
done
