set -ex
export TMPDIR=/tmp
mkdir -p gpurun_out/r2q
rocprofv3 -L > gpurun_out/r2q/counters.txt 2>&1 || true
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_BUSY_CYCLES -d gpurun_out/r2q/pmc1 -o run --output-format csv -- python3 tools/trace_stats.py 512 100 1200 > gpurun_out/r2q/pmc1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_IFETCH SQ_INST_LEVEL_VMEM SQ_WAVES SQ_INSTS_BRANCH -d gpurun_out/r2q/pmc2 -o run --output-format csv -- python3 tools/trace_stats.py 512 100 1200 > gpurun_out/r2q/pmc2.log 2>&1 || true
ls gpurun_out/r2q/pmc1 gpurun_out/r2q/pmc2 || true
