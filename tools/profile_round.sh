#!/bin/bash
# Collects the round's judged profiles on the GPU box: kernel stats, the two PMC passes (separately, as the
# MI355X guide prescribes) and the plain bench line.  Usage: bash tools/profile_round.sh <out-dir-under-gpurun_out>
set -e
export TMPDIR=/tmp
O=${GRAFT_REPO_ROOT:-$(pwd)}/gpurun_out/${1:-prof}
mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/stats -o run --output-format csv -- python3 bench.py --warmup 0 --cpu-sample 0 > $O/stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/fetch -o run --output-format csv -- python3 bench.py --warmup 0 --cpu-sample 0 > $O/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/write -o run --output-format csv -- python3 bench.py --warmup 0 --cpu-sample 0 > $O/write.log 2>&1
python3 bench.py > $O/bench.json 2> $O/bench.err
find $O -name "*.csv" | head -20
