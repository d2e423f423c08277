#!/bin/bash
# Collects the round's judged profiles on the GPU box: kernel stats, the two PMC passes (separately, as the
# MI355X guide prescribes), the hashes of the kernel sources / library they were taken on, and the plain bench line.
# Usage: bash tools/profile_round.sh <out-dir-under-gpurun_out> [bench.py arguments, e.g. --config c4]
set -e
export TMPDIR=/tmp
O=${GRAFT_REPO_ROOT:-$(pwd)}/gpurun_out/${1:-prof}
shift || true
mkdir -p $O
LEAN="--steps 1 --warmup 0 --cpu-sample 0 --parity-genes 0 --no-also --no-end-to-end"
python3 -c "import json, bench; from degnorm_amd import _lib; print(json.dumps({'source_sha256': bench.source_hash(), 'lib_sha256': bench.file_sha256(_lib.LIB_PATH)}))" > $O/hashes.json
rocprofv3 --kernel-trace --stats -d $O/stats -o run --output-format csv -- python3 bench.py $LEAN "$@" > $O/stats.log 2>&1
echo "stats done"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/fetch -o run --output-format csv -- python3 bench.py $LEAN "$@" > $O/fetch.log 2>&1
echo "fetch done"
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/write -o run --output-format csv -- python3 bench.py $LEAN "$@" > $O/write.log 2>&1
echo "write done"
python3 bench.py --no-also "$@" > $O/bench.json 2> $O/bench.err
find $O -name "*.csv" | head -20
