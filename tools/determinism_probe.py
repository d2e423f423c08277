"""Are two runs of the same library bit-identical, and do two library builds agree bit for bit?
usage (GPU box): python tools/determinism_probe.py <n_genes> <variant> [<variant> ...]     ('tree' = the product library)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

if os.environ.get('DN_VARIANT'):
    from collections import OrderedDict
    from degnorm_amd import synth
    from degnorm_amd.nmf import GeneNMFOA
    n = int(sys.argv[1])
    cfg = synth.CONFIGS['c2']
    covs = [synth.synth_gene(cfg['seed'], g, cfg['p'], cfg['l_min'], cfg['l_max'])[0] for g in range(n)]
    reads = np.vstack([synth.read_counts_from_coverage(c) for c in covs])
    out = []
    for rep in range(2):
        m = GeneNMFOA(degnorm_iter=5, nmf_iter=100)
        m.fit(OrderedDict(('g%d' % k, c) for k, c in enumerate(covs)), reads)
        out.append((m.rho.copy(), m.scale_factors.copy(), np.stack([t[:, :8] for t in m.traces])))
    print('%s: two runs in one process: rho identical %s, scale factors identical %s, traces identical %s' % (
        os.environ['DN_VARIANT'], np.array_equal(out[0][0], out[1][0]), np.array_equal(out[0][1], out[1][1]), np.array_equal(out[0][2], out[1][2])), flush=True)
    np.savez('/tmp/dn_det_%s.npz' % os.environ['DN_VARIANT'], rho=out[0][0], scale=out[0][1], traces=out[0][2])
    sys.exit(0)

ref = None
for v in sys.argv[2:]:
    env = dict(os.environ, DN_VARIANT=v)
    if v != 'tree':
        env['DN_LIB_PATH'] = os.path.join(ROOT, 'build_variants', 'lib_%s.so' % v)
    subprocess.run([sys.executable, os.path.abspath(__file__), sys.argv[1]], env=env, check=True)
    d = np.load('/tmp/dn_det_%s.npz' % v)
    if ref is None:
        ref = d
    else:
        diff = np.abs(d['rho'] - ref['rho'])
        tr = np.any(d['traces'] != ref['traces'], axis=(0, 2))
        print('%s vs %s: rho identical %s (max |d| %.3e, rows differing %d), scale factors identical %s (max rel %.3e), genes with another trace %d' % (
            v, sys.argv[2], np.array_equal(d['rho'], ref['rho']), diff.max(), int((diff.max(axis=1) > 0).sum()),
            np.array_equal(d['scale'], ref['scale']), float(np.max(np.abs(d['scale'] - ref['scale']) / ref['scale'])), int(tr.sum())), flush=True)
