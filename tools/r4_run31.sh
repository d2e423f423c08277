mkdir -p gpurun_out/r4F
# the driver's multi-GPU command line, rehearsed on the one GPU there is (gloo; the library's communicator falls back as a whole)
timeout -k 10 500 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --backend gloo --steps 2 --warmup 1 --cpu-sample 0 > gpurun_out/r4F/torchrun2.json 2> gpurun_out/r4F/torchrun2.err; echo "torchrun 2 ranks rc=$?"; tail -2 gpurun_out/r4F/torchrun2.err
python3 -c "
import json
d=json.load(open('gpurun_out/r4F/torchrun2.json')); print(d['value'], d['n_gpus'], d['ms_per_step'], d['scaling'], d['parity'].get('ok'), d.get('rccl'))"
timeout -k 10 500 python3 bench.py --gpus 2 --backend gloo --sharded-api --redeal --steps 2 --warmup 1 --cpu-sample 0 > gpurun_out/r4F/rehearsal_c2_2ranks.json 2> gpurun_out/r4F/rehearsal.err; echo "self-launched rehearsal rc=$?"
python3 -c "
import json
d=json.load(open('gpurun_out/r4F/rehearsal_c2_2ranks.json')); print(d['value'], d['n_gpus'], d['parity'].get('ok'), (d.get('end_to_end_sharded') or {}).get('total_s'))"
