set -ex
export TMPDIR=/tmp
mkdir -p gpurun_out/prof3
bash tools/profile_round.sh prof3_c2
mkdir -p gpurun_out/prof3_c4 && rocprofv3 --kernel-trace --stats -d gpurun_out/prof3_c4/stats -o run --output-format csv -- python3 bench.py --config c4 --warmup 0 --cpu-sample 0 --parity-genes 0 > gpurun_out/prof3_c4/stats.log 2>&1
python3 bench.py --config c4 > gpurun_out/prof3_c4/bench.json 2> gpurun_out/prof3_c4/bench.err
NCCL_DEBUG=VERSION python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 2 --warmup 1 --cpu-sample 0 > gpurun_out/prof3/torchrun_nccl_world1.log 2>&1
tail -2 gpurun_out/prof3/torchrun_nccl_world1.log | cut -c1-300
