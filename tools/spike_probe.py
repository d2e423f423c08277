import sys, time, os
sys.path.insert(0, os.getcwd())
import numpy as np
from degnorm_amd import synth
from degnorm_amd.nmf_mpi import ShardedNMFOA
n=50000; cfg=synth.CONFIGS['c4']
packed, lengths, reads, _ = synth.synth_packed(cfg['seed'], range(n), cfg['p'], cfg['l_min'], cfg['l_max'], n_threads=16)
eng = ShardedNMFOA(degnorm_iter=5, nmf_iter=100, downsample_rate=500)
eng.load_packed(packed, lengths, cfg['p'], reads)
dev=eng.dev
for rep in range(8):
    t=[time.perf_counter()]
    dev.ratio_svd_sums(fetch=False); t.append(time.perf_counter())
    pv=dev.init_partials(); t.append(time.perf_counter())
    dev.outer_begin_scaled(np.ones(50), 5); t.append(time.perf_counter())
    if rep%2==1:
        dev.fetch_outer(); t.append(time.perf_counter())
    print(rep, ' '.join('%.2f'%((b-a)*1e3) for a,b in zip(t[:-1],t[1:])), 'init kernel %.2f'%dev.last_init_ms(), flush=True)
