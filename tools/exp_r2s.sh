set -ex
mkdir -p gpurun_out/r2s
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 2 --steps 1 --warmup 1 --cpu-sample 0 --backend gloo > gpurun_out/r2s/rehearsal_2ranks.log 2>&1
tail -1 gpurun_out/r2s/rehearsal_2ranks.log | cut -c1-700
python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29542 bench.py --gpus 4 --steps 1 --warmup 1 --cpu-sample 0 --backend gloo > gpurun_out/r2s/rehearsal_4ranks.log 2>&1
tail -1 gpurun_out/r2s/rehearsal_4ranks.log | cut -c1-400
