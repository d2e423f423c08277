"""Where the wall time of a bench step goes outside the class kernels: timers around the host calls of ShardedNMFOA."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from degnorm_amd import synth, _lib
from degnorm_amd.nmf_mpi import ShardedNMFOA

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
cfg = dict(synth.CONFIGS['c2'])
packed, lengths, reads, _ = synth.synth_packed(cfg['seed'], range(n), cfg['p'], cfg['l_min'], cfg['l_max'])
eng = ShardedNMFOA(degnorm_iter=5, nmf_iter=100)
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
for nm in ('ratio_svd_sums', 'outer_begin', 'baseline_iteration', 'outer_partials', 'outer_apply', 'fetch_rows'):
    if hasattr(eng.dev, nm):
        timed(eng.dev, nm)
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
    print('step %d: %.1f ms = initialize %.1f + iterations %.1f (sum of kernel spans %.1f)' % (rep, (t2 - t0) * 1e3, (t1 - t0) * 1e3, (t2 - t1) * 1e3, sum(spans)))
    print('   ' + '  '.join('%s %s' % (k, ' '.join('%.1f' % x for x in v)) for k, v in acc.items()))
