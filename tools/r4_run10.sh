cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
bash tools/profile_round.sh r4prof_c2 > gpurun_out/r4prof_c2.log 2>&1; echo "c2 prof rc=$?"; tail -3 gpurun_out/r4prof_c2.log
O=gpurun_out/r4prof_c4; mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/stats -o run --output-format csv -- python3 bench.py --config c4 --steps 2 --warmup 1 --cpu-sample 0 --parity-genes 0 > $O/stats.log 2>&1; echo "c4 stats rc=$?"
python3 bench.py --config c4 > $O/bench.json 2> $O/bench.err; echo "c4 bench rc=$?"
ls $O/stats | head
