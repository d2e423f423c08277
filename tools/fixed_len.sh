for L in 3200 4000 5000; do echo "L $L"; timeout -k 10 200 python tools/trace_stats.py 1500 100 $L 2>&1 | grep -E "launch ms|per inner iteration:|pass cycles" | tail -3; done
