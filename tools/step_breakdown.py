"""Where the wall time of a bench step goes outside the class kernels: timers around the host calls of ShardedNMFOA.
usage: python tools/step_breakdown.py [n_genes] [c2|c4] [rccl] [rows]      rccl: the all-reduce through a one-rank RCCL process group (as bench.py);
rows: keep the raw DI rows of 160 genes per iteration (bench.py's parity sample)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from degnorm_amd import synth, _lib
from degnorm_amd.nmf_mpi import ShardedNMFOA

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
name = sys.argv[2] if len(sys.argv) > 2 else 'c2'
cfg = dict(synth.CONFIGS[name])
rate = 500 if name == 'c4' else 1
packed, lengths, reads, _ = synth.synth_packed(cfg['seed'], range(n), cfg['p'], cfg['l_min'], cfg['l_max'], n_threads=16)
comm = None
if 'rccl' in sys.argv:
    import torch, torch.distributed as dist
    from degnorm_amd.nmf_mpi import TorchComm
    torch.cuda.set_device(0)
    dist.init_process_group('nccl', init_method='tcp://127.0.0.1:29517', world_size=1, rank=0, device_id=torch.device('cuda', 0))
    comm = TorchComm(device='cuda:0')
    timed_comm = True
eng = ShardedNMFOA(comm=comm, degnorm_iter=5, nmf_iter=100, downsample_rate=rate)
if 'rows' in sys.argv:
    eng.history_rows = np.arange(0, n, max(1, n // 160))[:160]
eng.load_packed(packed, lengths, cfg['p'], reads)
acc = {}
def timed(obj, name):
    f = getattr(obj, name)
    def g(*a, **k):
        t0 = time.perf_counter()
        r = f(*a, **k)
        acc.setdefault(name, []).append((time.perf_counter() - t0) * 1e3)
        return r
    setattr(obj, name, g)
for nm in ('ratio_svd_sums', 'init_partials', 'outer_begin', 'outer_begin_scaled', 'baseline_iteration', 'outer_partials', 'outer_apply', 'fetch_rows', 'fetch_outer'):
    if hasattr(eng.dev, nm):
        timed(eng.dev, nm)
timed(eng, '_offsets')
if comm is not None:
    timed(comm, 'allreduce_sum')
timed(eng, '_warn_unconverged')
for rep in range(4):
    acc.clear()
    t0 = time.perf_counter()
    eng.initialize()
    t1 = time.perf_counter()
    spans = []
    for i in range(5):
        eng.iterate(i)
        spans.append(eng.span_ms[-1])
    t2 = time.perf_counter()
    eng.fetch_state()
    t3 = time.perf_counter()
    print('step %d: %.1f ms = initialize %.1f (kernel %.1f) + iterations %.1f (sum of kernel spans %.1f) + fetch_state %.1f' % (
        rep, (t3 - t0) * 1e3, (t1 - t0) * 1e3, eng.dev.last_init_ms(), (t2 - t1) * 1e3, sum(spans), (t3 - t2) * 1e3))
    print('   ' + '  '.join('%s %s' % (k, ' '.join('%.2f' % x for x in v)) for k, v in acc.items()))
