"""Diagnostic build only (DN_DEFINES=DN_STAMP=1 on dn_generic_nt256): cycles of the three phases of gen::k_ratio_svd_mg per gene
(pass 1 = Gram matrix on the matrix cores, solve = p x p power iteration, pass 2 = clamped row sums), on a config-4 slice.
usage: DN_LIB_PATH=build_variants/lib_<tag>.so python tools/init_phases.py [n_genes]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from degnorm_amd import synth, _lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8000
cfg = synth.CONFIGS['c4']
packed, lengths, reads, _ = synth.synth_packed(cfg['seed'], range(n), cfg['p'], cfg['l_min'], cfg['l_max'], n_threads=16)
dev = _lib.Device(0)
dev.hint_downsample(500)
dev.upload_packed(packed, lengths, cfg['p'])
for rep in range(3):
    est, cov, st = dev.ratio_svd_sums()
ms = dev.last_init_ms()
c = est[:, :4]
tot = c[:, :3].sum()
L = np.asarray(lengths, dtype=float)
print('kernel %.2f ms for %d genes (%.2f GB): %.0f us per gene per workgroup at 512 resident workgroups' % (ms, n, packed.nbytes / 1e9, ms * 1e3 * 512 / n))
print('cycles per gene (100 MHz s_memtime ticks x 24 = 2.4 GHz cycles): pass1 %.0f  solve %.0f  pass2 %.0f ; shares %.2f %.2f %.2f ; solver steps %.1f' % (
    c[:, 0].mean(), c[:, 1].mean(), c[:, 2].mean(), c[:, 0].sum() / tot, c[:, 1].sum() / tot, c[:, 2].sum() / tot, c[:, 3].mean()))
print('sum of phase ticks / (512 workgroups) = %.2f ms at 100 MHz' % (tot / 512 / 1e5))
if est.shape[1] >= 10:
    names = ('pass 1 main loop', 'partial group', 'tile sums', 'offsets', 'mirror', 'between genes (results, queue, gene record)')
    print('inside pass 1 and between the genes, cycles per gene: ' + '  '.join('%s %.0f' % (nm, est[:, 4 + k].mean()) for k, nm in enumerate(names)))
    A = np.stack([np.ones(n), L], axis=1)
    coef = np.linalg.lstsq(A, est[:, 4], rcond=None)[0]
    print('pass 1 main loop ticks = %.0f + %.3f x L' % (coef[0], coef[1]))
for name, col in (('pass1', 0), ('pass2', 2)):
    A = np.stack([np.ones(n), L], axis=1)
    coef = np.linalg.lstsq(A, c[:, col], rcond=None)[0]
    print('%s ticks = %.0f + %.3f x L' % (name, coef[0], coef[1]))
