"""Cost of the per-iteration collective (TorchComm.allreduce_sum of 3p+1 float64) under torchrun; run on the GPU box:
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29512 tools/allreduce_cost.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.distributed as dist
from degnorm_amd.nmf_mpi import TorchComm

lr = int(os.environ.get('LOCAL_RANK', 0))
torch.cuda.set_device(lr)
dist.init_process_group('nccl', device_id=torch.device('cuda', lr))
comm = TorchComm(device='cuda:{0}'.format(lr))
v = np.arange(31, dtype=np.float64)
for _ in range(5):
    comm.allreduce_sum(v)
torch.cuda.synchronize()
t0 = time.time()
for _ in range(200):
    out = comm.allreduce_sum(v)
dt = (time.time() - t0) / 200
if comm.rank == 0:
    print('allreduce_sum of 31 float64: %.1f us per call (world %d)' % (dt * 1e6, comm.size), out[:3])
dist.destroy_process_group()
