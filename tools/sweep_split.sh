for s in 2200 2500 2715 3044 3400; do echo "split $s"; DN_SPLIT_LEN=$s timeout -k 10 200 python tools/trace_stats.py 4000 100 2>&1 | grep -E "launch ms|per inner iteration:" ; done
