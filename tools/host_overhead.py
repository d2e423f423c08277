"""Where a step's wall time goes besides the kernels (matters for strong scaling: at 8 GPUs a step is ~280 ms).
Usage on the GPU box: python tools/host_overhead.py [n_genes]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from degnorm_amd import synth
from degnorm_amd.nmf_mpi import ShardedNMFOA

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2500
cfg = dict(synth.CONFIGS['c2'])
packed, lengths, reads, _ = synth.synth_packed(cfg['seed'], range(n), cfg['p'], cfg['l_min'], cfg['l_max'])
eng = ShardedNMFOA(degnorm_iter=5, nmf_iter=100)
eng.load_packed(packed, lengths, cfg['p'], reads)
for rep in range(2):
    t0 = time.perf_counter()
    eng.initialize()
    t_init = time.perf_counter() - t0
    rows = []
    for i in range(5):
        t1 = time.perf_counter()
        eng.iterate(i)
        wall = (time.perf_counter() - t1) * 1e3
        k0, k1 = eng.class_ms[-1][0], max(eng.class_ms[-1][1:])
        rows.append((wall, k0, k1))
    tot = (time.perf_counter() - t0) * 1e3
    print('rep %d: step %.1f ms, init %.2f ms; per iteration wall / wide kernel / narrow launch-to-end:' % (rep, tot, t_init * 1e3))
    for w, k0, k1 in rows:
        print('   %.2f  %.2f  %.2f   -> outside the kernels %.2f ms' % (w, k0, k1, w - max(k0, k1)))
