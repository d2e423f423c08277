mkdir -p gpurun_out/r4n
python3 tools/full_chain_parity.py 20000 gpurun_out/r4n/full_chain_parity_c2.json > gpurun_out/r4n/full_chain.log 2>&1; echo "rc=$?"; tail -12 gpurun_out/r4n/full_chain.log
