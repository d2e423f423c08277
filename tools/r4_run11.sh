mkdir -p gpurun_out/r4k
python3 tools/rows_latency.py profiles/round4/rows_inner_iteration_cycles.json > gpurun_out/r4k/rows_latency.txt 2>&1; cp profiles/round4/rows_inner_iteration_cycles.json gpurun_out/r4k/ 2>/dev/null; tail -3 gpurun_out/r4k/rows_latency.txt
for s in 41 42 43; do python3 tools/fuzz_parity.py --rounds 160 --seed $s --out gpurun_out/r4k/fuzz_seed$s.txt > gpurun_out/r4k/fuzz_seed$s.log 2>&1; echo "fuzz seed $s rc=$?"; tail -4 gpurun_out/r4k/fuzz_seed$s.log; done
python3 tools/fuzz_parity.py --chain --rounds 60 --seed 44 --out gpurun_out/r4k/fuzz_chain44.txt > gpurun_out/r4k/fuzz_chain44.log 2>&1; echo "fuzz chain rc=$?"; tail -4 gpurun_out/r4k/fuzz_chain44.log
