"""Diagnostics: run one baseline-selection launch on a config-2 draw and print per-gene counter statistics."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from degnorm_amd import synth, _lib

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
T = int(sys.argv[2]) if len(sys.argv) > 2 else 100
cfg = dict(synth.CONFIGS['c2'])
if len(sys.argv) > 3:
    cfg['l_min'] = cfg['l_max'] = int(sys.argv[3])
packed, lengths, reads, cls = synth.synth_packed(cfg['seed'], range(n), cfg['p'], cfg['l_min'], cfg['l_max'])
dev = _lib.Device(0)
dev.upload_packed(packed, lengths, cfg['p'])
est, cov, st = dev.ratio_svd_sums()
scale = np.ones(cfg['p'])
for rep in range(2):
    t0 = time.time()
    rho, flags, tr = dev.baseline_iteration(scale, nmf_iter=T)
    wall = (time.time() - t0) * 1e3
    print('launch ms', dev.last_kernel_ms(), ' class0 %.1f ms (%s)  class1 %.1f ms (%s)  split %d  iteration wall %.1f ms' % (
        dev.class_kernel_ms(0), dev.class_kernel_name(0), dev.class_kernel_ms(1), dev.class_kernel_name(1), dev.split_length(), wall))
calls = tr[:, 1].astype(float); cols = tr[:, 2].astype(float); steps = tr[:, 7].astype(float)
solves = calls * (T + 1)
print('genes', n, 'calls/gene', calls.mean(), 'sum cols/gene', cols.mean(), 'mean n per call', cols.sum() / calls.sum())
print('power steps per solve', steps.sum() / solves.sum(), 'max per gene avg', (steps[solves > 0] / solves[solves > 0]).max())
print('inner iterations total', solves.sum(), ' col-iters total %.3e' % (cols.sum() * T))
ms = dev.last_kernel_ms()
print('ns per inner iteration per CU (256 CUs): %.1f' % (ms * 1e6 * 256 / solves.sum()))
n0 = tr[:, 0]
print('n0 quantiles', np.percentile(n0, [10, 50, 90, 99]), 'frac cols beyond 2813:', np.maximum(n0 - 2813, 0).sum() / n0.sum())
print('exit codes', np.bincount(tr[:, 3], minlength=7), 'loop reasons', np.bincount(tr[:, 4], minlength=6))
if tr[:, 40:44].any():
    st = tr[:, 40:44].astype(float).sum(axis=0) * 1024
    print('stamps (cycles): pass %.3e  reduce %.3e  eigen %.3e  gene total %.3e' % tuple(st))
    print('per inner iteration: pass %.0f reduce %.0f eigen %.0f ; gene total per inner it %.0f' % tuple(st / (calls * T).sum()))
    print('pass cycles per column-per-lane: %.0f' % (st[0] / (cols.sum() * T / 256.)))
if tr[:, 40:44].any():
    # per-size-class cost of the pass inside the mixed workload
    ok = (calls > 0)
    per_col = tr[:, 40].astype(float) * 1024 / np.maximum(cols * T / 256., 1)
    per_it = (tr[:, 41] + tr[:, 42]).astype(float) * 1024 / np.maximum(calls * T, 1)
    edges = [0, 500, 1000, 1500, 2000, 2524, 3000, 4000, 6000]
    print('n0 class: genes, pass ticks per column-per-lane, (reduce+eigen) ticks per inner iteration, share of kernel time')
    tot = tr[:, 43].astype(float).sum()
    for a, b in zip(edges[:-1], edges[1:]):
        m = ok & (n0 >= a) & (n0 < b)
        if m.any():
            print('  [%4d,%4d): %5d  %7.0f  %7.0f   %.3f' % (a, b, m.sum(), np.median(per_col[m]), np.median(per_it[m]), tr[m, 43].sum() / tot))
if tr[:, 44:47].any():
    ext = tr[:, 44:47].astype(float).sum(axis=0) * 1024
    tot = tr[:, 43].astype(float).sum() * 1024
    inner = tr[:, 40:43].astype(float).sum() * 1024
    print('share of gene time: pass+reduce+eigen %.3f | nmf() calls %.3f (cold start %.3f, final pass %.3f, save/restore+rest %.3f) | outside nmf() %.3f'
          % (inner / tot, ext[0] / tot, ext[2] / tot, ext[1] / tot, (ext[0] - inner - ext[1] - ext[2]) / tot, 1 - ext[0] / tot))
    if tr[:, 47].any():
        print('candidate scan + compaction: %.3f of gene time; between nmf() calls %.3f; after the last call %.3f' % (
            tr[:, 47].astype(float).sum() * 1024 / tot, tr[:, 39].astype(float).sum() * 1024 / tot, tr[:, 38].astype(float).sum() * 1024 / tot))
    # utilisation of the resident workgroups: sum of per-gene cycles / (slots x kernel wall)
    for c in (0, 1):
        ms = dev.class_kernel_ms(c)
        if ms <= 0:
            continue
        m = (lengths > dev.split_length()) if c == 0 else (lengths <= dev.split_length())
        slots = 256 if c == 0 else 512
        busy = tr[m, 43].astype(float).sum() * 1024
        print('class %d: %d genes, sum of gene cycles %.3e = %.1f ms x %d slots at 2.4 GHz (kernel %.1f ms)' % (c, m.sum(), busy, busy / slots / 2.4e6, slots, ms))
