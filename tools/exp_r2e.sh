set -ex
mkdir -p gpurun_out/r2e
bash tools/sweep_split.sh 3600 4000 4400 5001 > gpurun_out/r2e/sweep.log 2>&1
cat gpurun_out/r2e/sweep.log
