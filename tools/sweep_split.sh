for s in 1600 2047 2400 2800; do echo "split $s"; DN_SPLIT_LEN=$s timeout -k 10 200 python tools/trace_stats.py 4000 100 2>&1 | grep -E "launch ms" | tail -1; done
