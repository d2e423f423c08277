"""Diagnostics: per-class kernel times of a baseline-selection sweep on the config-2 draw (after warm-up sweeps).
Usage: python tools/class_times.py [n_genes [l_min l_max]]   (DN_TINY_LEN / DN_SPLIT_LEN override the class boundaries)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from degnorm_amd import synth, _lib

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
cfg = dict(synth.CONFIGS['c2'])
if len(sys.argv) > 3:
    cfg['l_min'], cfg['l_max'] = int(sys.argv[2]), int(sys.argv[3])
packed, lengths, reads, cls = synth.synth_packed(cfg['seed'], range(n), cfg['p'], cfg['l_min'], cfg['l_max'])
dev = _lib.Device(0)
dev.upload_packed(packed, lengths, cfg['p'])
dev.ratio_svd_sums()
scale = np.ones(cfg['p'])
L = np.asarray(lengths)
for rep in range(3):
    rho, flags, tr = dev.baseline_iteration(scale, nmf_iter=100)
    split, tiny = dev.split_length(), dev.tiny_length()
    print('sweep %d: span %.1f ms | ' % (rep, dev.last_span_ms()) + ' | '.join(
        'class %d %s %.1f ms' % (c, dev.class_kernel_name(c), dev.class_kernel_ms(c)) for c in range(3)) +
        ' | split %d tiny %d genes %d / %d / %d' % (split, tiny, (L > split).sum(), ((L <= split) & (L > tiny)).sum(), (L <= tiny).sum()))
print('checksum rho %.12e flags %d calls %d' % (float(np.nansum(rho)), int(flags.sum()), int(tr[:, 1].sum())))
