"""Throughput in BASELINE config 4's regime (p = 50, take-every 500, L ~ U[501, 5000]; run-time-p kernels) on a slice of
the configuration: python tools/config4_speed.py [n_genes]   (the full configuration has 50 000 genes = 27.5 GB of fp32)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from degnorm_amd import synth
from degnorm_amd.nmf_mpi import ShardedNMFOA

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
cfg = dict(synth.CONFIGS['c4'])
if len(sys.argv) > 2:
    cfg['p'] = int(sys.argv[2])                 # other cohort sizes in the same down-sampled regime
t0 = time.time()
packed, lengths, reads, _ = synth.synth_packed(cfg['seed'], range(n), cfg['p'], cfg['l_min'], cfg['l_max'], n_threads=16)
print('synthetic slice: %d genes x %d samples, %.2f GB fp32, generated in %.1f s' % (n, cfg['p'], packed.nbytes / 1e9, time.time() - t0))
eng = ShardedNMFOA(degnorm_iter=5, nmf_iter=100, downsample_rate=cfg.get('downsample_rate', 500))
t0 = time.time()
eng.load_packed(packed, lengths, cfg['p'], reads)
print('upload %.2f s' % (time.time() - t0))
for rep in range(3):
    t0 = time.time()
    eng.initialize()
    t_init = time.time() - t0
    walls = []
    for i in range(5):
        t1 = time.time()
        eng.iterate(i)
        walls.append((time.time() - t1) * 1e3)
    dt = time.time() - t0
    print('run %d: %.3f s (init %.1f ms) for 5 outer iterations -> %.0f genes/s; per iteration wall ms: %s; kernel ms: %s' % (
        rep, dt, t_init * 1e3, n / dt, ', '.join('%.1f' % w for w in walls), ', '.join('%.1f' % k for k in eng.kernel_ms)))
