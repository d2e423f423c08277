# N > 1 code path of bench.py end to end on ONE GPU (every rank on device 0, gloo): not a scaling measurement
set -ex
mkdir -p gpurun_out/$1
for n in 2 4; do
python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 2954$n bench.py --gpus $n --backend gloo --steps 1 --warmup 1 --cpu-sample 0 > gpurun_out/$1/rehearsal_gloo_${n}ranks_one_gpu.log 2>&1
tail -1 gpurun_out/$1/rehearsal_gloo_${n}ranks_one_gpu.log | cut -c1-400
done
