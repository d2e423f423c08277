set -ex
mkdir -p gpurun_out/r2p
for ph in 1 2 3 4; do echo "== phase $ph"; DN_LIB_PATH=build_variants/lib_ph$ph.so timeout -k 10 200 python tools/trace_stats.py 512 100 1200 2>&1 | grep -E "per inner iteration:|stamps" | tail -2; done > gpurun_out/r2p/phases.log 2>&1
cat gpurun_out/r2p/phases.log
