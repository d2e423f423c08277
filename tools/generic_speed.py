"""Timing of the run-time-p kernel family (p > 12) on full-length genes."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from degnorm_amd import synth, _lib
p = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = int(sys.argv[2]) if len(sys.argv) > 2 else 512
packed, lengths, reads, cls = synth.synth_packed(3, range(n), p, 200, 3000)
dev = _lib.Device(0)
dev.upload_packed(packed, lengths, p)
dev.ratio_svd_sums()
dt = 1e9
for rep in range(3):                      # best of three: the first launch pays one-time set-up
    t0 = time.time()
    rho, flags, tr = dev.baseline_iteration(np.ones(p), nmf_iter=100)
    dt = min(dt, time.time() - t0)
print('p=%d genes=%d: %.2f s per outer iteration (%.1f genes/s), kernel %s, mean calls %.1f, mean power steps per solve %.1f' % (
    p, n, dt, n / dt, dev.class_kernel_name(0), tr[:, 1].mean(), tr[:, 7].sum() / max(1, (tr[:, 1] * 101).sum())))
