#!/bin/bash
# How much the pass of a long gene slows down when the whole chip runs long genes ("The memory side
# matters after all").  Needs the diagnostic build: DN_DEFINES=DN_STAMP=1 python -m degnorm_amd.build --force
for n in 8 64 256 1024; do
  echo "genes $n, L 5000"
  timeout -k 10 200 python tools/trace_stats.py $n 100 5000 2>&1 | grep -E "per inner iteration:|pass cycles"
done
for n in 8 256; do
  echo "genes $n, L 1800"
  timeout -k 10 200 python tools/trace_stats.py $n 100 1800 2>&1 | grep -E "per inner iteration:|pass cycles"
done
