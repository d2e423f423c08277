"""Throughput per outer iteration over the sample count p on config-2-shaped genes (L ~ U[200, 5000]): where the register tier
(8 <= p <= 12) and the class split stop, and what a 6-sample cohort (BASELINE configs[4]) gets.
usage (GPU box): python tools/p_sweep.py [n_genes] p [p ...]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from degnorm_amd import synth, _lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
for p in [int(a) for a in sys.argv[2:]] or [4, 6, 8, 10, 12, 14, 16]:
    packed, lengths, reads, cls = synth.synth_packed(2, range(n), p, 200, 5000, n_threads=16)
    dev = _lib.Device(0)
    dev.upload_packed(packed, lengths, p)
    dev.ratio_svd_sums()
    dt = 1e9
    for rep in range(3):
        t0 = time.time()
        rho, flags, tr = dev.baseline_iteration(np.ones(p), nmf_iter=100)
        dt = min(dt, time.time() - t0)
    cols = float(tr[:, 2].astype(np.float64).sum()) * 100
    flop = cols * (p * p + 9.0 * p)
    print('p=%2d genes=%d: %.3f s per outer iteration = %.0f genes/s; %.2f TFLOP/s of fp64 column work (%.3f of the vector peak); split %d pair %d; class ms %s' % (
        p, n, dt, n / dt, flop / dt / 1e12, flop / dt / 78.6e12, dev.split_length(), dev.tiny_length(),
        ' / '.join('%.1f' % dev.class_kernel_ms(c) for c in range(3))), flush=True)
    dev.close()
