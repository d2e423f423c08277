"""Diagnostics (DN_STAMP build): least-squares split of the pass time into a fixed part and a cost per column of each tier
(register / LDS / spill), per gene class, from the per-gene cycle stamps of one launch on a config-2 draw."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from degnorm_amd import synth, _lib

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
T = 100
cfg = dict(synth.CONFIGS['c2'])
packed, lengths, reads, cls = synth.synth_packed(cfg['seed'], range(n), cfg['p'], cfg['l_min'], cfg['l_max'])
dev = _lib.Device(0)
dev.upload_packed(packed, lengths, cfg['p'])
dev.ratio_svd_sums()
scale = np.ones(cfg['p'])
for rep in range(2):
    rho, flags, tr = dev.baseline_iteration(scale, nmf_iter=T)
split = dev.split_length()
calls = tr[:, 1].astype(float); cols = tr[:, 2].astype(float)
ok = calls > 0
L = np.asarray(lengths)
p = cfg['p']
for name, NT, sel in (('narrow (128 threads)', 128, ok & (L <= split)), ('wide (256 threads)', 256, ok & (L > split))):
    if sel.sum() < 10:
        continue
    nm = cols[sel] / calls[sel]                       # mean active columns per call
    rt = 10 * NT                                      # register tier (packed counts): 10 columns per lane at p = 10
    lds = (1950 if NT == 256 else 975)
    cr = np.minimum(nm, rt) / NT
    cl = np.clip(nm - rt, 0, lds) / NT
    cs = np.maximum(nm - rt - lds, 0) / NT
    y = tr[sel, 40].astype(float) * 1024 / (calls[sel] * T)          # pass cycles per inner iteration
    A = np.stack([np.ones_like(cr), cr, cl, cs], axis=1)
    coef, res, rk, sv = np.linalg.lstsq(A, y, rcond=None)
    pred = A @ coef
    print('%s: %d genes; pass cycles per inner iteration = %.0f + %.0f x reg-cols + %.0f x lds-cols + %.0f x spill-cols per lane  (rms residual %.0f of mean %.0f)'
          % (name, sel.sum(), coef[0], coef[1], coef[2], coef[3], np.sqrt(np.mean((y - pred) ** 2)), y.mean()))
    # raw view: genes binned by their mean active width (in columns per lane), median cycles per inner iteration
    cpl = nm / NT
    edges = np.arange(0, cpl.max() + 1.0, 1.0)
    row = []
    for a, b in zip(edges[:-1], edges[1:]):
        m = (cpl >= a) & (cpl < b)
        if m.sum() >= 5:
            row.append('%g-%g:%.0f' % (a, b, np.median(y[m])))
    print('   by columns per lane: ' + '  '.join(row))
    fx = (tr[sel, 41] + tr[sel, 42]).astype(float) * 1024 / (calls[sel] * T)
    print('   reduce + eigen per inner iteration: median %.0f' % np.median(fx))
