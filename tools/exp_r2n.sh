set -ex
mkdir -p gpurun_out/r2n
bash tools/sweep_split.sh 2600 3000 3400 3800 4200 > gpurun_out/r2n/sweep.log 2>&1
cat gpurun_out/r2n/sweep.log
